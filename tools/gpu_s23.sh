#!/bin/bash
# session 23: structural prefetch list (commit 0305dc8's Python, build/head_tree) vs the planner, same box
out=$PWD/gpurun_out/s23
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
run() { # label dir env...
  label=$1; dir=$2; shift; shift
  ( cd $dir && env "$@" timeout 600 python bench.py --no-fp16 --no-cpu-baseline --no-roofline --no-batch8 --no-dropin --steps 50 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$label bs1', d['ms_per_step'])" >> $out/bench.txt )
}
run off . MIXDQ_PREFETCH=0
run structural build/head_tree MIXDQ_PREFETCH=1
run planner . MIXDQ_PREFETCH=1
run structural build/head_tree MIXDQ_PREFETCH=1
run planner . MIXDQ_PREFETCH=1
run planner_own_only . MIXDQ_PREFETCH_MB=30 MIXDQ_PREFETCH_LEAD=0
run off . MIXDQ_PREFETCH=0
cat $out/bench.txt
