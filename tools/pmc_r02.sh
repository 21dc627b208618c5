#!/bin/bash
# Round-2 counter passes of the kernels that dominate the benchmarked graph: FETCH_SIZE and
# WRITE_SIZE in separate passes (MI355X_MICROARCH.md, HBM section) and one SQ pass (MFMA-busy,
# CU-busy, wait / issue-stall split).  Run on the GPU box from the repo root:
#   bash tools/pmc_r02.sh        -> gpurun_out/r02_pmc/{summary.json, *.csv}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r02_pmc
mkdir -p $out
SQ="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES"
i=0
while read M N K CFGS; do
  i=$((i+1))
  for pass in FETCH_SIZE WRITE_SIZE SQ; do
    ctr=$pass; [ $pass = SQ ] && ctr="$SQ"
    timeout 180 rocprofv3 --pmc $ctr --output-format csv -d $out/raw/${pass}_$i -o r -- python3 tools/pmc_gemm_probe.py $M $N $K $CFGS > $out/raw_${pass}_$i.log 2>&1
  done
  echo "$i $M $N $K $CFGS" >> $out/shapes.txt
done <<'SHAPES'
1024 1280 1280 56,42,41
1024 10240 1280 25
1024 1280 5120 45,41
1024 3840 1280 41
4096 640 640 35
8192 10240 1280 13,20
8192 1280 5120 25
SHAPES
python3 - <<'PY'
import csv, glob, collections, json, re
out = "gpurun_out/r02_pmc"
shapes = {l.split()[0]: l.split()[1:] for l in open(out + "/shapes.txt")}
res = {}
for i, (M, N, K, cfgs) in shapes.items():
    M, N, K = int(M), int(N), int(K)
    entry = {}
    for p in ("FETCH_SIZE", "WRITE_SIZE", "SQ"):
        f = glob.glob(f"{out}/raw/{p}_{i}/**/*counter_collection.csv", recursive=True)
        if not f:
            continue
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f[0])):
            if "igemm_kernel" not in r["Kernel_Name"]:
                continue
            k = re.sub(r"\(.*$", "", r["Kernel_Name"].replace("mixdq::(anonymous namespace)::", "").replace("void ", ""))
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            agg[k]["_dur_ns"].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        for k, d in agg.items():
            e = entry.setdefault(k, {})
            for c, v in d.items():
                v = v[1:] or v                       # first launch: cold instruction cache
                e[(p + ":" if c == "_dur_ns" else "") + c] = sum(v) / len(v)
    for k, e in entry.items():
        alg = M * K + N * K + 2 * M * N
        fk, wk = e.get("FETCH_SIZE", 0.0), e.get("WRITE_SIZE", 0.0)
        e["shape"] = f"M{M} N{N} K{K}"
        e["algorithmic_bytes"] = alg
        e["hbm_bytes_per_launch"] = int((2 * fk + wk) * 1024)   # gfx950: FETCH_SIZE counts half
        e["traffic_over_algorithmic"] = e["hbm_bytes_per_launch"] / alg
        if "SQ_VALU_MFMA_BUSY_CYCLES" in e and e.get("SQ_BUSY_CU_CYCLES"):
            # MFMA-pipe busy cycles are counted per SIMD (4 per CU), CU-busy cycles per CU
            e["mfma_util"] = e["SQ_VALU_MFMA_BUSY_CYCLES"] / (4.0 * e["SQ_BUSY_CU_CYCLES"])
        if "SQ_WAVE_CYCLES" in e and e["SQ_WAVE_CYCLES"]:
            for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
                if c in e:
                    e[c + "_frac_of_wave_cycles"] = e[c] / e["SQ_WAVE_CYCLES"]
        e["int8_tops_profiled"] = 2.0 * M * N * K / (e.get("SQ:_dur_ns", e.get("FETCH_SIZE:_dur_ns", 1)) * 1e-9) / 1e12
        res[f"{k} @ M{M} N{N} K{K}"] = e
json.dump(res, open(out + "/summary.json", "w"), indent=1, sort_keys=True)
for k, e in sorted(res.items()):
    print(k[:95], "| us", round(e.get("SQ:_dur_ns", 0) / 1e3, 1), "| hbm/alg", round(e["traffic_over_algorithmic"], 2),
          "| mfma util", round(e.get("mfma_util", 0), 3), "| wait", round(e.get("SQ_WAIT_ANY_frac_of_wave_cycles", 0), 2))
PY
find $out/raw -name "*counter_collection.csv" | while read f; do cp $f $out/$(echo $f | sed 's|.*/raw/||; s|/.*||')_counter_collection.csv; done
rm -rf $out/raw
