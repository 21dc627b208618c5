#!/bin/bash
# session 1 of round 4: kernel-boundary probe under runtime knobs, its trace, GEMM baselines
out=gpurun_out/s1
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
timeout 900 python tools/floor_probe.py --sweep > $out/floor_sweep.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o fp -- python3 tools/floor_probe.py > $out/floor_prof.txt 2>&1
python3 - <<PY > $out/floor_trace_summary.txt 2>&1
import csv, glob, collections
f = glob.glob("$out/prof/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
agg = collections.defaultdict(list)
for r in rows:
    agg[r["Kernel_Name"][:90]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in sorted(agg.items(), key=lambda kv: -len(kv[1]))[:12]:
    v.sort()
    print(len(v), "median us %.2f" % (v[len(v)//2] / 1e3), "min %.2f" % (v[0] / 1e3), k)
PY
rm -rf $out/prof
timeout 600 python tools/bench_gemm.py --bs 8 --cfgs 13,25,35,70 2>&1 | grep -v amdgpu > $out/w8_bs8.txt
timeout 600 python tools/bench_gemm.py --bs 1 --cfgs 25,35,41,45,56 2>&1 | grep -v amdgpu > $out/w8_bs1.txt
tail -3 $out/floor_sweep.txt
