#!/bin/bash
# session 15: short-key attention kernel A/B inside one library (MIXDQ_ATTN_SHORT=0/1)
out=gpurun_out/s15
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
for bs in 8 2 1; do
  for cfg in 1 4; do
    echo "== bs $bs cfg $cfg" >> $out/attn.txt
    timeout 300 python tools/bench_attn.py --bs $bs --impl hip --cfg $cfg --only cross 2>&1 | grep cross >> $out/attn.txt
  done
done
for v in 1 0 1 0; do
  MIXDQ_ATTN_SHORT=$v timeout 600 python bench.py --no-fp16 --no-cpu-baseline --no-roofline --steps 20 --batch 8 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('short=$v bs8', d['ms_per_step'])" >> $out/bench.txt
done
for v in 1 0; do
  MIXDQ_ATTN_SHORT=$v timeout 600 python bench.py --no-fp16 --no-cpu-baseline --no-roofline --steps 20 --batch 2 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('short=$v bs2', d['ms_per_step'])" >> $out/bench.txt
done
cat $out/attn.txt $out/bench.txt
