#!/bin/bash
# round 5, lease B: kernel statistics of the DROP-IN step (module swap alone: torch glue, no producer fusions)
out=gpurun_out/r05_b
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
for on in 0 1; do
  MIXDQ_F16IN=$on timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof$on -o bench -- python3 bench.py --no-fuse --no-fp16 --no-cpu-baseline --no-roofline --no-batch8 --no-dropin --steps 20 > $out/bench_dropin_f16in$on.json 2> $out/bench_dropin_f16in$on.err
  cp $(ls $out/prof$on/*/*kernel_stats.csv $out/prof$on/*kernel_stats.csv 2>/dev/null | head -1) $out/dropin_kernel_stats_f16in$on.csv 2>/dev/null
  rm -rf $out/prof$on
  tail -1 $out/bench_dropin_f16in$on.json | cut -c1-300
  head -40 $out/dropin_kernel_stats_f16in$on.csv | cut -c1-200
done
