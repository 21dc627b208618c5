#!/bin/bash
# Round 6, attention XCD map: tests, per-launch A/B (MIXDQ_ATTN_XCD=0/1 in ONE library), FETCH_SIZE counters, whole step.
#   -> gpurun_out/r06_attn_xcd/
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
out=gpurun_out/r06_attn_xcd
rm -rf $out; mkdir -p $out
timeout 900 python -m pytest tests/test_attention_gpu.py tests/test_fused_gpu.py tests/test_f16in_gpu.py -x -q -m gpu \
  -k "attention or qlinear_ln or traces or attn" > $out/pytest.txt 2>&1
tail -3 $out/pytest.txt
for bs in 1 8; do for x in 0 1 0 1; do
  echo "== bs $bs MIXDQ_ATTN_XCD=$x" >> $out/bench_attn.txt
  MIXDQ_ATTN_XCD=$x timeout 300 python tools/bench_attn.py --bs $bs --impl hip >> $out/bench_attn.txt 2>&1
done; done
cat $out/bench_attn.txt
# FETCH_SIZE per launch (x2 on gfx950), map off / on
for x in 0 1; do for spec in "1 4096 640" "1 1024 1280" "8 4096 640" "8 1024 1280"; do
  export MIXDQ_ATTN_XCD=$x
  d=$out/raw/x${x}_$(echo $spec | tr ' ' '_')
  timeout 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $d -o r -- python3 tools/pmc_gemm_probe.py attn $spec > $out/raw_x${x}.log 2>&1
  f=$(find $d -name "*counter_collection.csv" | head -1)
  python3 - "$f" "$x" "$spec" <<'PY' | tee -a $out/fetch.txt
import csv, sys
f, x, spec = sys.argv[1:4]
B, T, C = (int(v) for v in spec.split())
vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "attn" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE"]
alg = 4 * B * T * C * 2 / 1e6            # q, k, v read + o written (int8 o: the probe quantizes -> 3.5x; fp16 sizes as r5)
mb = 2 * sum(vals) / len(vals) * 1024 / 1e6 if vals else float("nan")   # FETCH_SIZE is in KiB; x2 (MI355X_MICROARCH.md HBM)
print(f"xcd_map={x} attn B={B} T={T} C={C}: FETCH_SIZE x2 = {mb:.1f} MB per launch over {len(vals)} launches; q+k+v = {3*B*T*C*2/1e6:.1f} MB")
PY
done; done
unset MIXDQ_ATTN_XCD
cp $(find $out/raw -name "*counter_collection.csv") $out/ 2>/dev/null
rm -rf $out/raw
for rep in 1 2; do for x in 0 1; do
  MIXDQ_ATTN_XCD=$x timeout 900 python bench.py --no-fp16 --no-cpu-baseline --no-roofline --no-dropin --no-lnchain --steps 40 > $out/bench_x${x}_$rep.json 2> $out/bench_x${x}_$rep.err
  python3 - $out/bench_x${x}_$rep.json $x $rep <<'PY' | tee -a $out/step_ab.txt
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("xcd_map", sys.argv[2], "rep", sys.argv[3], "ms %.3f" % d["ms_per_step"], "batch8 %.2f" % (d.get("batch8") or {}).get("ms_per_step", 0))
except Exception as e:
    print(sys.argv[2], "ERR", e)
PY
done; done
