#!/bin/bash
out=gpurun_out/r03_l
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
( time timeout 1200 python -m pytest tests/test_fused_gpu.py tests/test_ops_gpu.py tests/test_modules_gpu.py -q -x ) > $out/pytest.log 2>&1
tail -6 $out/pytest.log
S=$PWD/build/stamp/libmixdq_stamp.so
for args in "1024 10240 1280 --geglu" "8192 10240 1280 --geglu --cfg 18"; do
  echo "== $args"; MIXDQ_HIP_LIB=$S timeout 300 python tools/stamp_report.py $args 2>&1 | grep -v amdgpu.ids | tail -2
done
for bs in 1 8; do timeout 600 python tools/bench_geglu.py $bs 2>&1 | grep -v amdgpu.ids; done
