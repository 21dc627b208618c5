#!/bin/bash
# round 5, lease A: parity of the quantize-in-prologue GEMM, its cost per layer, the drop-in step
out=gpurun_out/r05_a
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
( time timeout 900 python -m pytest tests/test_f16in_gpu.py tests/test_modules_gpu.py -x -q 2>&1 | tail -15 ) > $out/pytest_f16in.txt 2>&1
cat $out/pytest_f16in.txt
timeout 600 python tools/bench_f16in.py > $out/bench_f16in_bs1.txt 2>&1
timeout 600 python tools/bench_f16in.py --chain > $out/bench_f16in_bs1_chain.txt 2>&1
cat $out/bench_f16in_bs1.txt $out/bench_f16in_bs1_chain.txt
MIXDQ_F16IN=0 timeout 900 python bench.py --no-cpu-baseline --no-roofline --no-batch8 > $out/bench_f16in_off.json 2> $out/bench_off.err
timeout 900 python bench.py --no-cpu-baseline --no-roofline --no-batch8 > $out/bench_f16in_on.json 2> $out/bench_on.err
python - <<PY
import json
for f in ("bench_f16in_off", "bench_f16in_on"):
    try:
        d = json.loads(open("$out/%s.json" % f).read().strip().splitlines()[-1])
        print(f, 'ms %.3f' % d['ms_per_step'], 'fp16', d['fp16']['ms_per_step'], 'dropin', d.get('dropin_unfused_ms_per_step'), d.get('dropin_unfused_kernels_per_step'), 'kernels', d.get('kernels_per_step'))
    except Exception as e:
        print(f, 'ERR', e); print(open("$out/%s.err" % f.replace('bench_f16in_','bench_')).read()[-1500:])
PY
