#!/bin/bash
# session 17: short-key attention kernel, un-traced step A/B again (three alternations)
out=gpurun_out/s17
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
for v in 1 0 1 0 1 0; do
  MIXDQ_ATTN_SHORT=$v timeout 600 python bench.py --no-fp16 --no-cpu-baseline --no-roofline --steps 30 --batch 8 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('short=$v bs8', d['ms_per_step'])" >> $out/bench.txt
done
cat $out/bench.txt
