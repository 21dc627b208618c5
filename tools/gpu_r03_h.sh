#!/bin/bash
out=gpurun_out/r03_h
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
( time timeout 1200 python -m pytest tests/test_ops_gpu.py tests/test_large_gpu.py tests/test_modules_gpu.py -q -x ) > $out/pytest.log 2>&1
tail -12 $out/pytest.log
timeout 600 python tools/bench_gemm.py --conv --cfgs 44,45,25,35 > $out/conv_bs1.log 2>&1; grep -v amdgpu.ids $out/conv_bs1.log | cut -c1-260
timeout 600 python tools/bench_gemm.py --conv --bs 8 --cfgs 25,13,20 > $out/conv_bs8.log 2>&1; grep -v amdgpu.ids $out/conv_bs8.log | cut -c1-260
