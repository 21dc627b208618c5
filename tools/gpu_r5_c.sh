#!/bin/bash
out=gpurun_out/r05_c
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
( time timeout 1500 python -m pytest tests/test_f16in_gpu.py tests/test_modules_gpu.py tests/test_unet_gpu.py tests/test_unet_full_gpu.py -x -q 2>&1 | tail -15 ) > $out/pytest.txt 2>&1
cat $out/pytest.txt
timeout 600 python tools/bench_f16in.py --bs 8 --L 60 > $out/bench_f16in_bs8.txt 2>&1
grep '^{' $out/bench_f16in_bs8.txt
