#!/bin/bash
# session 16: kernel trace of the batch-8 step with the short-key attention kernel on / off
out=gpurun_out/s16
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
for v in 1 0; do
  export MIXDQ_ATTN_SHORT=$v
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof$v -o bench -- python3 bench.py --no-fp16 --no-cpu-baseline --no-roofline --no-batch8 --no-dropin --steps 20 --batch 8 > $out/bench_prof_$v.json 2> $out/bench_prof_$v.err
  python3 tools/step_breakdown.py $(ls $out/prof$v/*/*kernel_trace.csv $out/prof$v/*kernel_trace.csv 2>/dev/null | head -1) 45 > $out/step_breakdown_short$v.txt 2>&1
  rm -rf $out/prof$v
done
grep -h "attn\|^step" $out/step_breakdown_short1.txt $out/step_breakdown_short0.txt | cut -c1-150
