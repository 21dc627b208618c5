#!/bin/bash
out=gpurun_out/r03_q
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
for rep in 1 2; do
for rows in 1 2; do
  MIXDQ_LN_ROWS=$rows timeout 600 python bench.py --batch 8 --steps 20 --no-fp16 --no-cpu-baseline --no-roofline > $out/bs8_rows${rows}_$rep.json 2> $out/err.txt
done
done
python - <<PY
import json, glob
for f in sorted(glob.glob("$out/bs8_*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], '%.3f' % d['ms_per_step'])
PY
