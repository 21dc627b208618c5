#!/bin/bash
out=gpurun_out/r03_c
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
S=$PWD/build/stamp/libmixdq_stamp.so
{
for args in "1024 10240 1280 --geglu" "1024 10240 1280 --geglu --cold" "1024 10240 1280" "1024 1280 1280 --res" "1024 1280 1280 --res --cold" "1024 1280 5120 --res" "1024 3840 1280" "8192 10240 1280 --geglu"; do
  echo "== $args"
  MIXDQ_HIP_LIB=$S timeout 300 python tools/stamp_report.py $args 2>&1 | grep -v amdgpu.ids
done
} > $out/stamps.log
cat $out/stamps.log
( time timeout 1800 python -m pytest tests/test_f16_gpu.py tests/test_unet_full_gpu.py tests/test_unet_gpu.py -q ) > $out/pytest.log 2>&1
tail -15 $out/pytest.log
