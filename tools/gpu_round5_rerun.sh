#!/bin/bash
# The part of tools/gpu_round5.sh that depends on the Python-side defaults (re-run after a default changed;
# the counter passes and the batch >= 8 lines of the first lease stay valid: same kernel sources)
tag=${1:-r05_final2}
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
( time timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -8 ) > $out/pytest_gpu.txt 2>&1
timeout 1500 python bench.py > $out/bench_default.json 2> $out/bench_default.err
timeout 900 python bench.py --baseline-config 2 --no-fp16 --no-cpu-baseline > $out/bench_w4a8_mixed_bs1.json 2> $out/bench_cfg2.err
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof1 -o bench -- python3 bench.py --no-fp16 --no-cpu-baseline --no-roofline --no-batch8 --no-dropin --no-lnchain --steps 20 --batch 1 > $out/bench_prof_bs1.json 2> $out/bench_prof_bs1.err
python3 tools/step_breakdown.py $(ls $out/prof1/*/*kernel_trace.csv $out/prof1/*kernel_trace.csv 2>/dev/null | head -1) 45 > $out/step_breakdown_bs1.txt 2>&1
cp $(ls $out/prof1/*/*kernel_stats.csv $out/prof1/*kernel_stats.csv 2>/dev/null | head -1) $out/bench_kernel_stats_bs1.csv 2>/dev/null
rm -rf $out/prof1
python - <<PY
import json
d = json.loads(open("$out/bench_default.json").read().strip().splitlines()[-1])
r = d["roofline"]
print({k: d.get(k) for k in ("ms_per_step","speedup_vs_fp16","speedup_vs_fp16_like_for_like","speedup_vs_fp16_dropin","dropin_unfused_ms_per_step","dropin_unfused_kernels_per_step","kernels_per_step","ln_in_gemm")})
print({k: r.get(k) for k in ("kernel","frac","avg_launch_us","in_step_avg_launch_us","frac_in_step","in_step_launches","traffic","mfma_util")}, r.get("traffic_source"))
print("batch8", d["batch8"]["ms_per_step"], "cpu", d["cpu_baseline"]["value"])
PY
cat $out/pytest_gpu.txt; head -12 $out/step_breakdown_bs1.txt
