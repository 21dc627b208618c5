#!/bin/bash
# L2 hit / miss counters of single GEMM launches (tools/pmc_gemm_probe.py M N K cfgs):
#   bash tools/pmc_l2.sh M N K cfgs   -> gpurun_out/pmc_l2/*.csv + a per-kernel summary on stdout
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/pmc_l2
mkdir -p $out
timeout 180 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --output-format csv -d $out/raw -o r -- python3 tools/pmc_gemm_probe.py "$@" > $out/run.log 2>&1
f=$(ls $out/raw/*/*counter_collection.csv $out/raw/*counter_collection.csv 2>/dev/null | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if "igemm" not in k: continue
    agg[k[:120]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in agg.items():
    print(k)
    for n, v in c.items():
        print("   ", n, len(v), sum(v) / len(v))
PY
tail -3 $out/run.log
