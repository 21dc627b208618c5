#!/bin/bash
out=gpurun_out/r03_r
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
timeout 900 python bench.py --baseline-config 2 --no-fp16 --no-cpu-baseline > $out/bench_cfg2.json 2> $out/bench_cfg2.err
timeout 900 python bench.py --baseline-config 4 --forwards-per-image 20 --steps 20 --warmup 2 --no-fp16 --no-cpu-baseline --no-roofline > $out/bench_cfg4.json 2> $out/bench_cfg4.err
timeout 900 python bench.py --w-config weight/weight_4.00 --a-config act/act_8.00 --w4-kernel --no-fp16 --no-cpu-baseline --no-roofline > $out/bench_w4_act8.json 2> $out/bench_w4_act8.err
python - <<'PY'
import json
for f in ('cfg2','cfg4','w4_act8'):
    try:
        d=json.loads(open(f'gpurun_out/r03_r/bench_{f}.json').read().strip().splitlines()[-1])
        print(f, 'ms_per_step %.3f'%d['ms_per_step'], 'value %.2f'%d['value'], d['config'].get('accelerated_layers'), d['config'].get('w4_kernel_layers'), d['config'].get('graphs_cached'), d['config'].get('workload'))
    except Exception as e: print(f, 'ERR', e)
PY
tail -3 $out/bench_cfg4.err
