#!/bin/bash
# session 14: short-key attention kernel (cross-attention, 77 keys) -- tests, microbench --cfg 1 vs 4, step A/B
out=gpurun_out/s14
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
base=$PWD/build/ab_base/libmixdq_hip.so
( time timeout 900 python -m pytest tests/test_attention_gpu.py tests/test_fused_gpu.py -q -m gpu 2>&1 | tail -8 ) > $out/pytest.txt 2>&1
for bs in 8 2 1; do
  for cfg in 1 4 2; do
    echo "== bs $bs cfg $cfg" >> $out/attn.txt
    timeout 300 python tools/bench_attn.py --bs $bs --impl hip --cfg $cfg 2>&1 | grep cross >> $out/attn.txt
  done
done
for v in new base new base; do
  lib=; [ $v = base ] && lib=$base
  MIXDQ_HIP_LIB=$lib timeout 600 python bench.py --no-fp16 --no-cpu-baseline --no-roofline --steps 20 --batch 8 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v bs8', d['ms_per_step'])" >> $out/bench.txt
done
cat $out/pytest.txt $out/attn.txt $out/bench.txt
