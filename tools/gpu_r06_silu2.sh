#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
out=gpurun_out/r06_silu2
rm -rf $out; mkdir -p $out
for rep in 1 2 3; do for m in 0 -1; do for bs in 1 8; do
  e="MIXDQ_GN_SILU_TAB=$m"; [ $m = -1 ] && e="MIXDQ_UNUSED=1"
  env $e timeout 900 python bench.py --batch $bs --no-fp16 --no-cpu-baseline --no-roofline --no-dropin --no-lnchain --no-batch8 --steps 40 > $out/b.json 2> $out/b.err
  python3 - $out/b.json $m $rep $bs <<'PY' | tee -a $out/step_ab.txt
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("silu table", "off" if sys.argv[2] == "0" else "default (>= 2 Mi elements)", "rep", sys.argv[3], "batch", sys.argv[4], "ms %.3f" % d["ms_per_step"])
except Exception as e:
    print(sys.argv[2], "ERR", e)
PY
done; done; done
