#!/bin/bash
out=gpurun_out/r05_i
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
for rep in 1 2; do
for on in 0 1; do
  MIXDQ_LN_CHAIN=$on timeout 900 python bench.py --no-fp16 --no-cpu-baseline --no-roofline --no-batch8 --steps 40 > $out/bench_ln$on.$rep.json 2> $out/bench_ln$on.$rep.err
  python - <<PY
import json
try:
    d = json.loads(open("$out/bench_ln$on.$rep.json").read().strip().splitlines()[-1])
    print("LN_CHAIN=$on rep $rep ms %.3f kernels %s dropin %.2f" % (d["ms_per_step"], d.get("kernels_per_step"), d.get("dropin_unfused_ms_per_step", 0)))
except Exception as e:
    print("ERR", e); print(open("$out/bench_ln$on.$rep.err").read()[-1500:])
PY
done
done
for on in 0 1; do
  MIXDQ_LN_CHAIN=$on timeout 900 python bench.py --batch 2 --no-fp16 --no-cpu-baseline --no-roofline --no-batch8 --no-dropin --steps 20 > $out/bench_bs2_ln$on.json 2> $out/bench_bs2_ln$on.err
  python - <<PY
import json
d = json.loads(open("$out/bench_bs2_ln$on.json").read().strip().splitlines()[-1])
print("batch 2 LN_CHAIN=$on ms %.3f" % d["ms_per_step"])
PY
done
( time timeout 2400 python -m pytest tests -q -m gpu -x 2>&1 | tail -6 ) > $out/pytest_gpu.txt 2>&1
cat $out/pytest_gpu.txt
