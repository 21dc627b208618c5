#!/bin/bash
# One GPU lease of the round: full GPU test suite, default bench line, rocprofv3
# kernel trace of the graph that is benchmarked.  Outputs under gpurun_out/r02_*.
#   bash tools/gpu_round.sh [tag]
tag=${1:-a}
out=gpurun_out/r02_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $out/pytest.log 2>&1
tail -5 $out/pytest.log
timeout 900 python bench.py > $out/bench_default.json 2> $out/bench_default.err
tail -c 1500 $out/bench_default.json
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o bench -- python3 bench.py --no-fp16 --no-cpu-baseline --no-roofline --steps 20 > $out/bench_prof.json 2> $out/bench_prof.err
python3 tools/step_breakdown.py $(ls $out/prof/*/*kernel_trace.csv $out/prof/*kernel_trace.csv 2>/dev/null | head -1) 40 > $out/step_breakdown_bs1.txt 2>&1
cp $(ls $out/prof/*/*kernel_stats.csv $out/prof/*kernel_stats.csv 2>/dev/null | head -1) $out/bench_kernel_stats.csv 2>/dev/null
head -30 $out/step_breakdown_bs1.txt
# keep the merge-back small: the full trace is tens of MB
rm -rf $out/prof
