#!/bin/bash
# One lease for the round's record, on the FINAL tree: the GPU suite, the benchmark lines of the named
# configurations, the rocprofv3 kernel traces of the benchmarked graphs (batch 1 and 8), the counter passes.
#   bash tools/gpu_round6.sh <tag>     -> gpurun_out/<tag>/...   (copy what is to be judged into profiles/)
tag=${1:-r06_final}
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
( time timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -8 ) > $out/pytest_gpu.txt 2>&1
timeout 1500 python bench.py > $out/bench_default.json 2> $out/bench_default.err
timeout 900 python bench.py --batch 8 --steps 10 --no-cpu-baseline > $out/bench_bs8.json 2> $out/bench_bs8.err
timeout 900 python bench.py --baseline-config 4 --forwards-per-image 20 --steps 20 --warmup 2 --no-fp16 --no-cpu-baseline > $out/bench_bs16_cfg4_20steps.json 2> $out/bench_cfg4.err
timeout 900 python bench.py --baseline-config 2 --no-fp16 --no-cpu-baseline > $out/bench_w4a8_mixed_bs1.json 2> $out/bench_cfg2.err
timeout 1500 python bench.py --baseline-config 3 --gpus 1 --steps 4 --warmup 1 --no-fp16 --no-cpu-baseline > $out/bench_bs64_cfg3_1gpu.json 2> $out/bench_cfg3.err
MIXDQ_SHARE_DEVICE=1 MIXDQ_DIST_BACKEND=gloo timeout 600 python bench.py --gpus 2 --tiny --no-fp16 --no-cpu-baseline --no-roofline > $out/bench_gpus2_tiny_shared_device.json 2> $out/bench_gpus2.err
for bs in 1 8; do
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof$bs -o bench -- python3 bench.py --no-fp16 --no-cpu-baseline --no-roofline --no-batch8 --no-dropin --no-lnchain --steps 20 --batch $bs > $out/bench_prof_bs$bs.json 2> $out/bench_prof_bs$bs.err
  python3 tools/step_breakdown.py $(ls $out/prof$bs/*/*kernel_trace.csv $out/prof$bs/*kernel_trace.csv 2>/dev/null | head -1) 45 > $out/step_breakdown_bs$bs.txt 2>&1
  cp $(ls $out/prof$bs/*/*kernel_stats.csv $out/prof$bs/*kernel_stats.csv 2>/dev/null | head -1) $out/bench_kernel_stats_bs$bs.csv 2>/dev/null
  rm -rf $out/prof$bs
done
# the module swap with swap_glue=True, traced: which kernels the drop-in step runs now
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/profg -o bench -- python3 bench.py --no-fuse --swap-glue --no-fp16 --no-cpu-baseline --no-roofline --no-batch8 --no-dropin --no-lnchain --steps 20 > $out/bench_prof_dropin_glue.json 2> $out/bench_prof_dropin_glue.err
python3 tools/step_breakdown.py $(ls $out/profg/*/*kernel_trace.csv $out/profg/*kernel_trace.csv 2>/dev/null | head -1) 30 > $out/step_breakdown_dropin_glue.txt 2>&1
rm -rf $out/profg
timeout 2400 bash tools/pmc_r06.sh > $out/pmc.txt 2>&1
python - <<PY
import json, glob
for f in sorted(glob.glob("$out/bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        r = d.get('roofline') or {}
        print(f.split('/')[-1], 'n_gpus', d['n_gpus'], 'ms_per_step %.3f' % d['ms_per_step'], 'value %.2f' % d['value'],
              'vs fp16', d.get('speedup_vs_fp16'), 'like-for-like', d.get('speedup_vs_fp16_like_for_like'), 'dropin', d.get('speedup_vs_fp16_dropin'),
              'dropin_glue', d.get('speedup_vs_fp16_dropin_glue'), 'frac', r.get('frac'), 'in-step', r.get('frac_in_step'), 'whole', r.get('whole_step_frac'),
              'batch8', (d.get('batch8') or {}).get('ms_per_step'), 'dropin ms', d.get('dropin_unfused_ms_per_step'), d.get('dropin_glue_ms_per_step'),
              'kernels', d.get('kernels_per_step'))
    except Exception as e:
        print(f, 'ERR', e)
PY
# (appended late in round 6) what the statistics pass's pixels-in-flight are worth: one library, the switch, alternating
for rep in 1 2; do for u in 1 0; do
  MIXDQ_GN_STATS_UNROLL=$u timeout 300 python bench.py --no-fp16 --no-cpu-baseline --no-roofline --no-dropin --no-lnchain --no-batch8 --steps 40 > $out/ab_gnstats_${u}_$rep.json 2>/dev/null
  python3 -c "
import json,sys
try: print('MIXDQ_GN_STATS_UNROLL=$u rep $rep ms %.3f' % json.loads(open('$out/ab_gnstats_${u}_$rep.json').read().strip().splitlines()[-1])['ms_per_step'])
except Exception as e: print('ERR', e)" | tee -a $out/ab_gnstats.txt
done; done
cat $out/pytest_gpu.txt; tail -24 $out/pmc.txt; head -14 $out/step_breakdown_bs1.txt
