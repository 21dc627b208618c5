#!/bin/bash
# same-box A/B: round-2 library vs the current one, batch 1 and batch 8
out=gpurun_out/r03_m
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
for rep in 1; do
for lib in r02 cur; do
  L=$PWD/mixdq_amd/libmixdq_hip.so; [ $lib = r02 ] && L=$PWD/build/ab_r02/libmixdq_hip.so
  for bs in 1 8; do
    MIXDQ_HIP_LIB=$L timeout 900 python bench.py --no-fp16 --no-cpu-baseline --batch $bs --steps 20 > $out/bench_${lib}_bs${bs}_$rep.json 2> $out/bench_${lib}_bs${bs}_$rep.err
    python - $out/bench_${lib}_bs${bs}_$rep.json $lib $bs <<'PY'
import json, sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r=d['roofline']
print(sys.argv[2], 'bs', sys.argv[3], 'ms_per_step %.3f' % d['ms_per_step'], '| igemm ms %.3f' % r['all_igemm']['ms_per_step'], '| dominant', r['kernel'], '%.1f us' % r['avg_launch_us'])
PY
  done
done
done
( time timeout 2400 python -m pytest tests -m gpu -q -x ) > $out/pytest.log 2>&1
tail -5 $out/pytest.log
