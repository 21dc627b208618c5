#!/bin/bash
# Round 6: SiLU by LDS table in the GroupNorm apply pass -- tests, per-launch A/B (MIXDQ_GN_SILU_TAB=0 / auto), whole step
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
out=gpurun_out/r06_silu
rm -rf $out; mkdir -p $out
timeout 900 python -m pytest tests/test_fused_gpu.py -x -q -m gpu -k "silu or groupnorm" > $out/pytest.txt 2>&1
tail -5 $out/pytest.txt
for m in 0 1 0 1; do
  echo "== MIXDQ_GN_SILU_TAB=$m (1: forced for every launch with a finalize launch)" >> $out/bench_norms.txt
  MIXDQ_GN_SILU_TAB=$m timeout 300 python tools/bench_norms.py 2>&1 | grep "^GN" >> $out/bench_norms.txt
done
cat $out/bench_norms.txt
for rep in 1 2; do for m in 0 -1; do
  e="MIXDQ_GN_SILU_TAB=$m"; [ $m = -1 ] && e="MIXDQ_GN_SILU_TAB_MIN=8388608"
  env $e timeout 900 python bench.py --batch 8 --no-fp16 --no-cpu-baseline --no-roofline --no-dropin --no-lnchain --no-batch8 --steps 20 > $out/bench_bs8_m${m}_$rep.json 2> $out/bench_bs8_m${m}_$rep.err
  python3 - $out/bench_bs8_m${m}_$rep.json $m $rep <<'PY' | tee -a $out/step_ab.txt
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("silu table", "off" if sys.argv[2] == "0" else "auto", "rep", sys.argv[3], "batch 8 ms %.3f" % d["ms_per_step"])
except Exception as e:
    print(sys.argv[2], "ERR", e)
PY
done; done
