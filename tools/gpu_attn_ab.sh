#!/bin/bash
timeout 1200 python -m pytest tests/test_attention_gpu.py tests/test_fused_gpu.py -x -q -m gpu 2>&1 | tail -5
for v in base new base new; do
  lib=$PWD/mixdq_amd/libmixdq_hip.so; [ $v = base ] && lib=$PWD/build/ab/libmixdq_base.so
  echo "== $v"
  for bs in 1 8; do
    MIXDQ_HIP_LIB=$lib timeout 300 python tools/bench_attn.py --bs $bs --impl hip 2>&1 | grep -v amdgpu.ids | cut -c1-200
  done
done
