#!/bin/bash
# session 28: the lifted row limit as the default -- batch 8 / 16 lines, the full-size UNet identity tests
out=gpurun_out/s28
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
timeout 600 python bench.py --batch 8 --steps 10 --no-fp16 --no-cpu-baseline > $out/bench_bs8.json 2> $out/bench_bs8.err
timeout 600 python bench.py --baseline-config 4 --forwards-per-image 20 --steps 20 --warmup 2 --no-fp16 --no-cpu-baseline --no-roofline > $out/bench_bs16_cfg4_20steps.json 2> $out/bench_cfg4.err
( time timeout 900 python -m pytest tests/test_unet_full_gpu.py -q -m gpu -x 2>&1 | tail -3 ) > $out/pytest.txt 2>&1
python - <<PY
import json
for f in ("bench_bs8", "bench_bs16_cfg4_20steps"):
    d = json.loads(open("$out/%s.json" % f).read().strip().splitlines()[-1])
    print(f, d["ms_per_step"], d["value"])
PY
cat $out/pytest.txt
