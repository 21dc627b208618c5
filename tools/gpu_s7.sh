#!/bin/bash
# session 7: residual request behind the params->LDS stores, kernel-argument warm-up (A/B: build/ab_nowarm)
out=gpurun_out/s7
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
A=$PWD/build/ab_r04a/libmixdq_hip.so
N=$PWD/build/ab_nowarm/libmixdq_hip.so
( time timeout 1800 python -m pytest tests/test_ops_gpu.py tests/test_fused_gpu.py tests/test_large_gpu.py tests/test_modules_gpu.py tests/test_unet_full_gpu.py -q -m gpu -k "not over_4_gib and not shard_size" 2>&1 | tail -6 ) > $out/pytest.txt 2>&1
for v in A N C A N C; do
  lib=""; [ $v = A ] && lib=$A; [ $v = N ] && lib=$N
  MIXDQ_HIP_LIB=$lib timeout 300 python tools/floor_probe.py 2>&1 | grep "^{" | sed "s/^/$v /" >> $out/floor.txt
done
for v in A N C A N C; do
  lib=""; [ $v = A ] && lib=$A; [ $v = N ] && lib=$N
  MIXDQ_HIP_LIB=$lib timeout 600 python bench.py --no-fp16 --no-cpu-baseline --no-roofline --no-batch8 --no-dropin --steps 50 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v bs1', d['ms_per_step'])" >> $out/bench.txt
done
for v in A N C A C; do
  lib=""; [ $v = A ] && lib=$A; [ $v = N ] && lib=$N
  MIXDQ_HIP_LIB=$lib timeout 600 python bench.py --no-fp16 --no-cpu-baseline --no-roofline --steps 20 --batch 8 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v bs8', d['ms_per_step'])" >> $out/bench.txt
done
export MIXDQ_HIP_LIB=$PWD/build/stamp/libmixdq_stamp.so
for spec in "8192 1280 1280 --cfg 27 --res" "8192 1280 1280 --cfg 25 --res" "1024 1280 1280 --cfg 56 --res"; do
  echo "== $spec" >> $out/stamps.txt
  timeout 300 python tools/stamp_report.py $spec 2>&1 | grep -v amdgpu.ids | tail -2 >> $out/stamps.txt
done
unset MIXDQ_HIP_LIB
cat $out/pytest.txt $out/floor.txt $out/bench.txt $out/stamps.txt
