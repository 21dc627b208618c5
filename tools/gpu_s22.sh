#!/bin/bash
# session 22: prefetch planner with a bounded look-back -- batch 1 A/B over budget / lead
out=gpurun_out/s22
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
run() { # label batch env...
  label=$1; bs=$2; shift; shift
  env "$@" timeout 600 python bench.py --no-fp16 --no-cpu-baseline --no-roofline --no-batch8 --no-dropin --steps 50 --batch $bs 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$label bs$bs', d['ms_per_step'])" >> $out/bench.txt
}
run off 1 MIXDQ_PREFETCH=0
run mb48_lead4 1 MIXDQ_PREFETCH=1
run mb30_lead0 1 MIXDQ_PREFETCH_MB=30 MIXDQ_PREFETCH_LEAD=0
run mb48_lead1 1 MIXDQ_PREFETCH_LEAD=1
run mb48_lead2 1 MIXDQ_PREFETCH_LEAD=2
run mb64_lead4 1 MIXDQ_PREFETCH_MB=64
run mb36_lead4 1 MIXDQ_PREFETCH_MB=36
run mb48_lead4 1 MIXDQ_PREFETCH=1
run off 1 MIXDQ_PREFETCH=0
cat $out/bench.txt
