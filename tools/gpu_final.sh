#!/bin/bash
# end-of-round lease: the full round script, then the batch-8 and batch-16 bench lines
bash tools/gpu_round.sh ${1:-g}
out=gpurun_out/r02_${1:-g}
timeout 900 python bench.py --batch 8 --steps 10 --no-cpu-baseline > $out/bench_bs8.json 2> $out/bench_bs8.err
python3 -c "import json,sys; r=json.loads(open('$out/bench_bs8.json').read().strip().splitlines()[-1]); print('bs8', r['ms_per_step'], r['value'], r['fp16']['ms_per_step'], r['roofline'].get('kernel'), r['roofline']['frac'])"
timeout 900 python bench.py --baseline-config 4 --steps 10 --no-cpu-baseline --no-fp16 > $out/bench_bs16.json 2> $out/bench_bs16.err
python3 -c "import json,sys; r=json.loads(open('$out/bench_bs16.json').read().strip().splitlines()[-1]); print('bs16', r['ms_per_step'], r['value'])"
