#!/bin/bash
# session 4: DIRECT epilogue (64x80 tiles) + late residual batch (large tiles): parity, A/B vs the previous commit
out=gpurun_out/s4
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
A=$PWD/build/ab_r04a/libmixdq_hip.so
( time timeout 1500 python -m pytest tests/test_ops_gpu.py tests/test_fused_gpu.py tests/test_large_gpu.py tests/test_modules_gpu.py tests/test_f16_gpu.py tests/test_unet_full_gpu.py -q -m gpu -x -k "not shard_size and not over_4_gib" 2>&1 | tail -8 ) > $out/pytest.txt 2>&1
for v in A B A B; do
  lib=""; [ $v = A ] && lib=$A
  MIXDQ_HIP_LIB=$lib timeout 300 python tools/floor_probe.py 2>&1 | grep "^{" | sed "s/^/$v /" >> $out/floor.txt
done
for v in A B A B; do
  lib=""; [ $v = A ] && lib=$A
  MIXDQ_HIP_LIB=$lib timeout 600 python bench.py --no-fp16 --no-cpu-baseline --no-roofline --no-batch8 --no-dropin --steps 50 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v bs1', d['ms_per_step'])" >> $out/bench.txt
done
for v in A B A B; do
  lib=""; [ $v = A ] && lib=$A
  MIXDQ_HIP_LIB=$lib timeout 600 python bench.py --no-fp16 --no-cpu-baseline --no-roofline --steps 20 --batch 8 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v bs8', d['ms_per_step'])" >> $out/bench.txt
done
export MIXDQ_HIP_LIB=$PWD/build/stamp/libmixdq_stamp.so
for spec in "1024 1280 1280 --cfg 56 --res" "1024 1280 5120 --cfg 45 --res" "8192 1280 1280 --cfg 25 --res" "8192 1280 5120 --cfg 70 --res"; do
  echo "== $spec" >> $out/stamps.txt
  timeout 300 python tools/stamp_report.py $spec 2>&1 | grep -v amdgpu.ids | tail -2 >> $out/stamps.txt
done
unset MIXDQ_HIP_LIB
timeout 300 python tools/bench_gemm.py --bs 8 --cfgs 70,25 2>&1 | grep -v amdgpu | cut -c1-200 > $out/gemm_bs8.txt
cat $out/pytest.txt $out/floor.txt $out/bench.txt $out/stamps.txt
