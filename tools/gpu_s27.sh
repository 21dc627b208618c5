#!/bin/bash
# session 27: the prefetch at batch 8 (row limit lifted) -- measurement only
out=gpurun_out/s27
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
run() { label=$1; shift
  env "$@" timeout 600 python bench.py --no-fp16 --no-cpu-baseline --no-roofline --steps 20 --batch 8 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$label bs8', d['ms_per_step'])" >> $out/bench.txt
}
run default MIXDQ_PREFETCH=1
run rows_unlimited MIXDQ_PREFETCH_MAX_ROWS=1000000
run default MIXDQ_PREFETCH=1
run rows_unlimited MIXDQ_PREFETCH_MAX_ROWS=1000000
cat $out/bench.txt
