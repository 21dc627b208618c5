#!/bin/bash
# same-box A/B of bench.py under several environments, alternating:
#   bash tools/gpu_ab.sh <tag> "<env 1>" "<env 2>" ["<env 3>" ...] [-- bench.py flags]
# e.g.  bash tools/gpu_ab.sh r05_delay "MIXDQ_PREFETCH_DELAY=0" "MIXDQ_PREFETCH_DELAY=1" -- --steps 40
tag=${1:?tag}; shift
envs=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do envs+=("$1"); shift; done
[ "$1" = "--" ] && shift
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
for rep in 1 2 3; do i=0; for E in "${envs[@]}"; do i=$((i+1))
  env $E timeout 900 python bench.py --no-fp16 --no-cpu-baseline --no-roofline --no-dropin --no-lnchain "$@" > $out/bench_${i}_$rep.json 2> $out/bench_${i}_$rep.err
  python - $out/bench_${i}_$rep.json "$E" $rep <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("[%s] rep %s ms %.3f" % (sys.argv[2], sys.argv[3], d["ms_per_step"]), "batch8 %.2f" % (d.get("batch8") or {}).get("ms_per_step", 0))
except Exception as e:
    print(sys.argv[2], "ERR", e)
PY
done; done | tee $out/ab.txt
