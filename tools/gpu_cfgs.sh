#!/bin/bash
# bench lines of the other BASELINE.json configurations (profiles/r02_bench_*.json)
out=gpurun_out/r02_cfgs
mkdir -p $out
timeout 900 python bench.py --baseline-config 2 --no-cpu-baseline > $out/bench_w4a8_mixed_bs1.json 2> $out/w4.err
timeout 900 python bench.py --batch 8 --no-cpu-baseline --steps 10 > $out/bench_bs8.json 2> $out/bs8.err
timeout 900 python bench.py --baseline-config 4 --no-cpu-baseline --steps 10 > $out/bench_bs16_cfg4.json 2> $out/bs16.err
timeout 1500 python bench.py --baseline-config 3 --no-cpu-baseline --steps 6 --warmup 2 > $out/bench_bs64_cfg3_1gpu.json 2> $out/bs64.err
for f in $out/*.json; do python3 -c "
import json,sys
d=json.load(open('$f'))
print('$f'.split('/')[-1], round(d['ms_per_step'],2), 'ms', round(d['value'],1), d['unit'], 'fp16', round(d['fp16']['ms_per_step'],1), 'x', round(d['speedup_vs_fp16'],2), d['scaling'], d['config']['workload'], d['roofline']['kernel'], round(d['roofline']['frac'],3), d['memory'].get('saving_vs_fp16'))
"; done
