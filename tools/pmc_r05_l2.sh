#!/bin/bash
# Round 5, VERDICT r4 #2: does the batch-8 GEMM+GEGLU launch's HBM-side over-fetch (2.88 x the algorithmic bytes on
# the 256x256 four-phase tile) move with the shape of the patch an XCD's co-resident tiles cover, and does the TIME
# follow it?  The super-row height GM of the tile map (MIXDQ_IGEMM_GM) sets that shape: an XCD runs 160 of the
# 32 x 40 tiles, 32 at a time = GM m-tiles x (32 / GM) n-tiles (GM <= 32).  Counters in separate passes
# (MI355X_MICROARCH.md, HBM section: FETCH_SIZE x 2 on gfx950).
#   bash tools/pmc_r05_l2.sh   -> gpurun_out/r05_l2/{l2_shape.txt, *_counter_collection.csv}
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
out=gpurun_out/r05_l2
rm -rf $out; mkdir -p $out
SQ="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES"
i=0
for spec in "geglu 8192 10240 1280" "lin 8192 3840 1280"; do
for gm in 1 2 4 8 16 32; do
  i=$((i+1))
  export MIXDQ_IGEMM_GM=$gm
  for pass in FETCH_SIZE WRITE_SIZE SQ; do
    ctr=$pass; [ $pass = SQ ] && ctr="$SQ"
    timeout 240 rocprofv3 --pmc $ctr --output-format csv -d $out/raw/${pass}_$i -o r -- python3 tools/pmc_gemm_probe.py $spec 70 > $out/raw_${pass}_$i.log 2>&1
  done
  echo "$i $spec" >> $out/shapes.txt
  echo "$i GM=$gm" >> $out/gm.txt
done
done
unset MIXDQ_IGEMM_GM
python3 tools/pmc_r03_summary.py $out > $out/summary_print.txt 2>&1
python3 - <<PY
import json
res = json.load(open("$out/summary.json"))
PY
# the summary keys collide across GM values (same kernel @ same shape): redo per index
python3 - <<'PY'
import collections, csv, glob, re
out = "gpurun_out/r05_l2"
gms = dict(l.split() for l in open(out + "/gm.txt"))
shapes = {l.split()[0]: l.split()[1:] for l in open(out + "/shapes.txt")}
lines = ["# (8192, 10240, 1280) GEMM+GEGLU and (8192, 3840, 1280) Linear on the 256x256x128 four-phase tile (cfg 70), batch 8:",
         "# tile-map super-row height GM -> HBM-side bytes per launch (FETCH_SIZE x 2 + WRITE_SIZE), over the algorithmic bytes,",
         "# MFMA-busy = SQ_VALU_MFMA_BUSY_CYCLES / (4 SQ_BUSY_CU_CYCLES), us per launch under the SQ pass",
         "launch GM co-resident_patch(m x n tiles) fetch_MB write_MB hbm_over_algorithmic mfma_util us"]
for i in sorted(shapes, key=int):
    kind, M, N, K = shapes[i][0], *map(int, shapes[i][1:4])
    alg = M * K + N * K + (M * N // 2 if kind == "geglu" else 2 * M * N)
    vals = {}
    for p in ("FETCH_SIZE", "WRITE_SIZE", "SQ"):
        f = glob.glob(f"{out}/raw/{p}_{i}/**/*counter_collection.csv", recursive=True)
        if not f:
            continue
        agg = collections.defaultdict(list)
        dur = []
        for r in csv.DictReader(open(f[0])):
            if "igemm_kernel" not in r["Kernel_Name"]:
                continue
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
            if r["Counter_Name"] == "SQ_BUSY_CU_CYCLES":
                dur.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        for c, v in agg.items():
            v = v[1:] or v
            vals[c] = sum(v) / len(v)
        if dur:
            d = dur[1:] or dur
            vals["us"] = sum(d) / len(d) / 1e3
    gm = int(gms[i].split("=")[1])
    fk, wk = vals.get("FETCH_SIZE", 0), vals.get("WRITE_SIZE", 0)
    hbm = (2 * fk + wk) * 1024
    util = vals.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (4 * vals.get("SQ_BUSY_CU_CYCLES", 1))
    lines.append(f"{kind}({M},{N},{K}) {gm} {min(gm, 32)}x{max(1, 32 // gm)} {2 * fk * 1024 / 1e6:.1f} {wk * 1024 / 1e6:.1f} {hbm / alg:.2f} {util:.3f} {vals.get('us', 0):.1f}")
open(out + "/l2_shape.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
find $out/raw -name "*counter_collection.csv" | while read f; do cp $f $out/$(echo $f | sed 's|.*/raw/||; s|/.*||')_counter_collection.csv; done
rm -rf $out/raw
