#!/bin/bash
out=gpurun_out/r02_final
mkdir -p $out
timeout 900 python bench.py --no-fp16 --no-cpu-baseline --steps 50 > $out/w8a8_bs1.json 2>/dev/null
timeout 900 python bench.py --baseline-config 2 --no-fp16 --no-cpu-baseline --steps 50 > $out/w4a8_mixed_bs1.json 2>/dev/null
timeout 900 python bench.py --w-config weight/weight_4.00 --a-config act/act_8.00 --w4-kernel --no-fp16 --no-cpu-baseline --steps 50 > $out/w4_act8_bs1.json 2>/dev/null
timeout 900 python bench.py --batch 8 --no-fp16 --no-cpu-baseline --steps 10 > $out/w8a8_bs8.json 2>/dev/null
timeout 900 python bench.py --batch 16 --no-fp16 --no-cpu-baseline --steps 10 > $out/w8a8_bs16.json 2>/dev/null
for f in $out/*.json; do python3 -c "
import json
d=json.load(open('$f'))
r=d['roofline']
print('$f'.split('/')[-1], round(d['ms_per_step'],2), 'ms', round(d['value'],1), 'img/s | accel', d['config']['accelerated_layers'], 'w4', d['config']['w4_kernel_layers'], '| static MB', round(list(d['memory'].values())[0]['static_mb']), '| dominant', r['kernel'], round(r['frac'],3), '| igemm', round(r['all_igemm']['ms_per_step'],2), 'ms', r['all_igemm']['launches_per_step'], 'launches', round(r['all_igemm']['int8_ops_per_step']/1e12,3), 'Top')
"; done
