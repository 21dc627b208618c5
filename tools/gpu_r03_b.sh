#!/bin/bash
# Round 3, lease B: in-kernel phase stamps of the dominant launches; attention + full-size UNet tests;
# GEMM+GEGLU tile choice at batch 1 / 8.
out=gpurun_out/r03_b
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
S=$PWD/build/stamp/libmixdq_stamp.so
{
for args in "1024 10240 1280 --geglu" "1024 10240 1280 --geglu --cold" "1024 10240 1280" "1024 1280 1280 --res" "1024 1280 1280 --res --cold" "1024 1280 5120 --res" "1024 3840 1280" "8192 10240 1280 --geglu" "8192 10240 1280 --geglu --cfg 70"; do
  echo "== $args"
  MIXDQ_HIP_LIB=$S timeout 300 python tools/stamp_report.py $args 2>&1 | grep -v amdgpu.ids
done
} > $out/stamps.log
cat $out/stamps.log
( time timeout 1800 python -m pytest tests/test_attention_gpu.py tests/test_unet_full_gpu.py tests/test_fused_gpu.py -q ) > $out/pytest.log 2>&1
tail -15 $out/pytest.log
for bs in 1 8; do timeout 600 python tools/bench_geglu.py $bs 2>&1 | grep -v amdgpu.ids; done > $out/geglu.log
cat $out/geglu.log
