#!/bin/bash
# FETCH_SIZE / WRITE_SIZE passes (separate, as the guide prescribes) for one GEMM shape under a few
# tile configurations: bash tools/pmc_traffic_gemm.sh M N K cfg,cfg   -> gpurun_out/pmc_traffic/
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 120 rocprofv3 --pmc $c --output-format csv -d gpurun_out/pmc_traffic/$c -o r -- python3 tools/pmc_gemm_probe.py $1 $2 $3 $4 > gpurun_out/pmc_traffic_$c.log 2>&1
done
python3 - <<'PY'
import csv, glob, collections, re
res = collections.defaultdict(dict)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"gpurun_out/pmc_traffic/{c}/*counter_collection.csv")
    if not f:
        print(c, "missing"); continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        if "igemm_kernel" in r["Kernel_Name"] and r["Counter_Name"] == c:
            agg[re.sub(r"\(.*$", "", r["Kernel_Name"].replace("mixdq::(anonymous namespace)::", ""))].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        v = v[1:] or v                     # first launch: cold instruction cache / L2
        res[k][c] = sum(v) / len(v)
for k, d in res.items():
    fk, wk = d.get("FETCH_SIZE", 0), d.get("WRITE_SIZE", 0)
    print(k, "FETCH_SIZE KiB", round(fk, 1), "WRITE_SIZE KiB", round(wk, 1), "-> HBM-side bytes per launch (2 x fetch + write):", int((2 * fk + wk) * 1024))
PY
