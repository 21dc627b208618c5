#!/bin/bash
# same-box A/B of two LIBRARIES under the same Python (tools/ab_build.sh <commit> base builds the other one):
#   bash tools/gpu_ab_lib.sh <tag> [bench.py flags]   -> base = build/ab_base/libmixdq_hip.so, new = this tree's
tag=${1:?tag}; shift
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
for rep in 1 2 3; do for v in base new; do
  lib=$PWD/mixdq_amd/libmixdq_hip.so; [ $v = base ] && lib=$PWD/build/ab_base/libmixdq_hip.so
  MIXDQ_HIP_LIB=$lib timeout 900 python bench.py --no-fp16 --no-cpu-baseline --no-roofline --no-dropin --no-lnchain --steps 40 "$@" > $out/bench_${v}_$rep.json 2> $out/bench_${v}_$rep.err
  python - $out/bench_${v}_$rep.json $v $rep <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], "rep", sys.argv[3], "ms %.3f" % d["ms_per_step"], "batch8 %.2f" % (d.get("batch8") or {}).get("ms_per_step", 0))
except Exception as e:
    print(sys.argv[2], "ERR", e)
PY
done; done | tee $out/ab.txt
