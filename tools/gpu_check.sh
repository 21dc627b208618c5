#!/bin/bash
# One lease: the whole GPU suite, then the batch-8 / batch-1 / batch-16 bench lines (no FP16 legs).
out=gpurun_out/check
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
( time timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -6 ) > $out/pytest.txt 2>&1
timeout 900 python bench.py --batch 8 --steps 10 --no-fp16 --no-cpu-baseline > $out/bench_bs8.json 2> $out/bench_bs8.err
timeout 900 python bench.py --no-fp16 --no-cpu-baseline > $out/bench_bs1.json 2> $out/bench_bs1.err
timeout 900 python bench.py --baseline-config 4 --forwards-per-image 20 --steps 20 --warmup 2 --no-fp16 --no-cpu-baseline --no-roofline > $out/bench_cfg4.json 2> $out/bench_cfg4.err
cat $out/pytest.txt
python - <<PY
import json, glob
for f in sorted(glob.glob("$out/bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        r = d.get('roofline') or {}
        print(f.split('/')[-1], 'ms_per_step %.3f' % d['ms_per_step'], 'setup %.0f s' % d['setup_s'], r.get('kernel'), r.get('avg_launch_us'), r.get('frac'), (r.get('all_igemm') or {}).get('ms_per_step'))
    except Exception as e:
        print(f, 'ERR', e)
PY
