#!/bin/bash
out=gpurun_out/r02_w4b
mkdir -p $out
timeout 900 python -m pytest tests/test_unet_gpu.py tests/test_ops_gpu.py -x -q -k "unet or w4" 2>&1 | tail -4
timeout 900 python bench.py --baseline-config 2 --no-fp16 --no-cpu-baseline --steps 100 > $out/bench_w4a8.json 2> $out/bench_w4a8.err
python3 -c "import json;d=json.load(open('$out/bench_w4a8.json'));print('w4a8', round(d['ms_per_step'],3), d['config']['accelerated_layers'], d['config']['w4_kernel_layers'], d['memory'], d['roofline']['all_igemm'])"
timeout 900 python bench.py --no-fp16 --no-cpu-baseline --steps 100 > $out/bench_w8a8.json 2> $out/bench_w8a8.err
python3 -c "import json;d=json.load(open('$out/bench_w8a8.json'));print('w8a8', round(d['ms_per_step'],3), d['memory'], d['roofline']['all_igemm'])"
