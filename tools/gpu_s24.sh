#!/bin/bash
# session 24: which weights are worth their bytes in the payload?  (MIXDQ_PREFETCH_SKIP_MB leaves larger tensors cold)
out=gpurun_out/s24
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
run() { # label env...
  label=$1; shift
  env "$@" timeout 600 python bench.py --no-fp16 --no-cpu-baseline --no-roofline --no-batch8 --no-dropin --steps 50 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$label bs1', d['ms_per_step'])" >> $out/bench.txt
}
run off MIXDQ_PREFETCH=0
run all MIXDQ_PREFETCH=1
run skip_over_10MB MIXDQ_PREFETCH_SKIP_MB=10
run skip_over_6MB MIXDQ_PREFETCH_SKIP_MB=6
run all MIXDQ_PREFETCH=1
run skip_over_10MB MIXDQ_PREFETCH_SKIP_MB=10
cat $out/bench.txt
