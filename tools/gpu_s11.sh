#!/bin/bash
# session 11: cfg 28 (128x320 on 16 waves of 32 x 80) against 27 / 25
out=gpurun_out/s11
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
( time timeout 1500 python -m pytest tests/test_ops_gpu.py -q -m gpu 2>&1 | tail -8 ) > $out/pytest.txt 2>&1
timeout 900 python tools/bench_gemm.py --bs 8 --cfgs 25,27,28 2>&1 | grep -v amdgpu | cut -c1-260 >> $out/gemm.txt
timeout 900 python tools/bench_gemm.py --bs 2 --cfgs 25,27,28 2>&1 | grep -v amdgpu | cut -c1-260 >> $out/gemm2.txt
cat $out/pytest.txt; cat $out/gemm.txt; echo == bs2; cat $out/gemm2.txt
