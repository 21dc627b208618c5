#!/bin/bash
out=gpurun_out/r05_f
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
( time timeout 600 python -m pytest tests/test_fused_gpu.py -x -q -k "layernorm or qlinear_ln" 2>&1 | tail -8 ) > $out/pytest_ln.txt 2>&1
cat $out/pytest_ln.txt
timeout 600 python tools/bench_ln_gemm.py > $out/bench_ln_gemm.txt 2>&1
grep '^{' $out/bench_ln_gemm.txt || tail -20 $out/bench_ln_gemm.txt
for rep in 1 2; do
for on in 0 1; do
  MIXDQ_LN_CHAIN=$on timeout 900 python bench.py --no-fp16 --no-cpu-baseline --no-roofline --no-batch8 --no-dropin --steps 40 > $out/bench_ln$on.$rep.json 2> $out/bench_ln$on.$rep.err
  python - <<PY
import json
try:
    d = json.loads(open("$out/bench_ln$on.$rep.json").read().strip().splitlines()[-1])
    print("LN_CHAIN=$on rep $rep ms %.3f" % d["ms_per_step"])
except Exception as e:
    print("ERR", e); print(open("$out/bench_ln$on.$rep.err").read()[-1500:])
PY
done
done
