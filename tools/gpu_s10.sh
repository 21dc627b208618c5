#!/bin/bash
# session 10: halo conv -- LDS swizzle for the 16-lane ds_read_b128 groups, 160-channel tile (93); A/B vs build/ab_base
out=gpurun_out/s10
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
base=$PWD/build/ab_base/libmixdq_hip.so
( time timeout 1500 python -m pytest tests/test_ops_gpu.py tests/test_large_gpu.py tests/test_modules_gpu.py tests/test_unet_gpu.py -q -m gpu -k "not over_4_gib and not shard_size" 2>&1 | tail -8 ) > $out/pytest.txt 2>&1
for bs in 8 1; do
  echo "== new bs$bs" >> $out/conv.txt
  timeout 900 python tools/bench_gemm.py --conv --bs $bs --cfgs 90,91,92,93 2>&1 | grep -v amdgpu | cut -c1-260 >> $out/conv.txt
  echo "== base bs$bs" >> $out/conv.txt
  MIXDQ_HIP_LIB=$base timeout 900 python tools/bench_gemm.py --conv --bs $bs --cfgs 90,91,92 2>&1 | grep -v amdgpu | cut -c1-260 >> $out/conv.txt
done
for v in new base new base; do
  lib=; [ $v = base ] && lib=$base
  MIXDQ_HIP_LIB=$lib timeout 600 python bench.py --no-fp16 --no-cpu-baseline --no-roofline --no-batch8 --no-dropin --steps 50 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v bs1', d['ms_per_step'])" >> $out/bench.txt
done
for v in new base new base; do
  lib=; [ $v = base ] && lib=$base
  MIXDQ_HIP_LIB=$lib timeout 600 python bench.py --no-fp16 --no-cpu-baseline --no-roofline --steps 20 --batch 8 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v bs8', d['ms_per_step'])" >> $out/bench.txt
done
cat $out/pytest.txt $out/bench.txt; cat $out/conv.txt
