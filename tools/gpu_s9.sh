#!/bin/bash
# session 9: K rotation (MIXDQ_IGEMM_KROT=0/1 A/B in one library)
out=gpurun_out/s9
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
( time timeout 1800 python -m pytest tests/test_ops_gpu.py tests/test_fused_gpu.py tests/test_large_gpu.py tests/test_modules_gpu.py tests/test_unet_full_gpu.py -q -m gpu -k "not over_4_gib and not shard_size" 2>&1 | tail -6 ) > $out/pytest.txt 2>&1
for v in 0 1 0 1; do
  MIXDQ_IGEMM_KROT=$v timeout 300 python tools/floor_probe.py 2>&1 | grep "^{" | sed "s/^/krot$v /" >> $out/floor.txt
done
for v in 0 1; do
  echo "== KROT=$v" >> $out/gemm.txt
  MIXDQ_IGEMM_KROT=$v timeout 600 python tools/bench_gemm.py --bs 8 --cfgs 70,27,35 2>&1 | grep -v amdgpu | cut -c1-230 >> $out/gemm.txt
  MIXDQ_IGEMM_KROT=$v timeout 600 python tools/bench_gemm.py --bs 1 --cfgs 27,35,45,56 2>&1 | grep -v amdgpu | cut -c1-230 >> $out/gemm.txt
  MIXDQ_IGEMM_KROT=$v timeout 600 python tools/bench_geglu_cfgs.py 2>&1 | grep -v amdgpu >> $out/gemm.txt
  MIXDQ_IGEMM_KROT=$v timeout 600 python tools/bench_geglu_cfgs.py --bs8 2>&1 | grep -v amdgpu >> $out/gemm.txt
done
for v in 0 1 0 1; do
  MIXDQ_IGEMM_KROT=$v timeout 600 python bench.py --no-fp16 --no-cpu-baseline --no-roofline --no-batch8 --no-dropin --steps 50 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('krot$v bs1', d['ms_per_step'])" >> $out/bench.txt
done
for v in 0 1 0 1; do
  MIXDQ_IGEMM_KROT=$v timeout 600 python bench.py --no-fp16 --no-cpu-baseline --no-roofline --steps 20 --batch 8 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('krot$v bs8', d['ms_per_step'])" >> $out/bench.txt
done
cat $out/pytest.txt $out/floor.txt $out/bench.txt; grep -E "==|M8192|M1024|M32768|M4096|geglu" $out/gemm.txt | cut -c1-200
