#!/bin/bash
out=gpurun_out/r02_w4
mkdir -p $out
timeout 900 python bench.py --baseline-config 2 --no-fp16 --no-cpu-baseline --steps 100 > $out/bench_w4a8.json 2> $out/bench_w4a8.err
python3 -c "import json;d=json.load(open('$out/bench_w4a8.json'));print('w4a8', round(d['ms_per_step'],3), d['config']['accelerated_layers'], d['config']['w4_kernel_layers'], d['memory'])"
timeout 900 python bench.py --no-fp16 --no-cpu-baseline --steps 100 > $out/bench_w8a8.json 2> $out/bench_w8a8.err
python3 -c "import json;d=json.load(open('$out/bench_w8a8.json'));print('w8a8', round(d['ms_per_step'],3), d['memory'])"
timeout 900 python bench.py --w-config weight/weight_4.00 --a-config act/act_8.00 --w4-kernel --no-fp16 --no-cpu-baseline --steps 100 > $out/bench_w4_a8.json 2> $out/bench_w4_a8.err
python3 -c "import json;d=json.load(open('$out/bench_w4_a8.json'));print('w4.00+act8.00', round(d['ms_per_step'],3), d['config']['accelerated_layers'], d['config']['w4_kernel_layers'])"
