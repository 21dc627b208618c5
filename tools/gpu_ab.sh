#!/bin/bash
# same-box A/B of bench.py under two environments:  bash tools/gpu_ab.sh <tag> "<envA>" "<envB>" [batch sizes]
tag=$1; A=$2; B=$3; shift 3; sizes=${@:-1 8}
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
for rep in 1 2; do for v in A B; do
  E=$A; [ $v = B ] && E=$B
  for bs in $sizes; do
    env $E timeout 900 python bench.py --no-fp16 --no-cpu-baseline --batch $bs --steps 20 > $out/bench_${v}_bs${bs}_$rep.json 2> $out/bench_${v}_bs${bs}_$rep.err
    python - $out/bench_${v}_bs${bs}_$rep.json "$v[$E]" $bs <<'PY'
import json, sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r=d['roofline']
print(sys.argv[2], 'bs', sys.argv[3], 'ms_per_step %.3f' % d['ms_per_step'], '| igemm ms %.3f' % r['all_igemm']['ms_per_step'], '| dominant', r['kernel'], '%.1f us' % r['avg_launch_us'])
PY
  done
done; done
