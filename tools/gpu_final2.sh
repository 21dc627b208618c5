#!/bin/bash
# closing lease of the round: suite + smoke, the remaining bench lines, MFMA-busy of the large tiles
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r02_h
mkdir -p $out
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $out/pytest.log 2>&1
grep -E "passed|failed" $out/pytest.log | tail -1
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python bench.py --batch 8 --steps 10 --no-cpu-baseline > $out/bench_bs8.json 2> $out/bench_bs8.err
python3 -c "import json; r=json.loads(open('$out/bench_bs8.json').read().strip().splitlines()[-1]); print('bs8', r['ms_per_step'], r['value'], r['roofline'].get('kernel'), r['roofline']['frac'])"
timeout 900 python bench.py --baseline-config 3 --gpus 1 --steps 5 --no-cpu-baseline --no-fp16 > $out/bench_bs64.json 2> $out/bench_bs64.err
python3 -c "import json; r=json.loads(open('$out/bench_bs64.json').read().strip().splitlines()[-1]); print('bs64', r['ms_per_step'], r['value'])"
timeout 900 python bench.py --baseline-config 2 --steps 20 --no-cpu-baseline --no-fp16 > $out/bench_w4a8.json 2> $out/bench_w4a8.err
python3 -c "import json; r=json.loads(open('$out/bench_w4a8.json').read().strip().splitlines()[-1]); print('w4a8', r['ms_per_step'], r['value'], r.get('memory',{}).get('w8a8'))"
timeout 180 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d $out/pmc -o r -- python3 tools/pmc_gemm_probe.py 8192 3840 1280 13,70 > $out/pmc.log 2>&1
python3 - $out <<'PY'
import csv, sys, glob, collections
f = glob.glob(sys.argv[1] + "/pmc/**/*counter_collection.csv", recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f[0])):
    if "igemm" in r["Kernel_Name"]:
        agg[r["Kernel_Name"][40:100]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in agg.items():
    m = {n: sum(v) / len(v) for n, v in c.items()}
    print(k, {n: round(v) for n, v in m.items()}, "mfma/busy_cu =", round(m["SQ_VALU_MFMA_BUSY_CYCLES"] / m["SQ_BUSY_CU_CYCLES"], 3))
PY
rm -rf $out/pmc/*/*.db
