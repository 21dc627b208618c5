#!/bin/bash
# rocprofv3 kernel trace of the benchmarked graph -> per-step breakdown by kernel.
#   bash tools/gpu_profile.sh <tag> [bench.py flags]     -> gpurun_out/<tag>/{step_breakdown.txt, bench_kernel_stats.csv}
tag=${1:?tag}; shift
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o bench -- python3 bench.py --no-fp16 --no-cpu-baseline --no-roofline --no-batch8 --no-dropin --no-lnchain --steps 20 "$@" > $out/bench_prof.json 2> $out/bench_prof.err
python3 tools/step_breakdown.py $(ls $out/prof/*/*kernel_trace.csv $out/prof/*kernel_trace.csv 2>/dev/null | head -1) 45 > $out/step_breakdown.txt 2>&1
cp $(ls $out/prof/*/*kernel_stats.csv $out/prof/*kernel_stats.csv 2>/dev/null | head -1) $out/bench_kernel_stats.csv 2>/dev/null
cat $out/step_breakdown.txt
rm -rf $out/prof
