#!/bin/bash
# session 21: prefetch planner (every weight behind the first self-attention launch) -- tests, batch 1 / 2 / 4 A/B
out=gpurun_out/s21
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
( time timeout 1500 python -m pytest tests/test_attention_gpu.py tests/test_unet_gpu.py tests/test_unet_full_gpu.py -q -m gpu 2>&1 | tail -8 ) > $out/pytest.txt 2>&1
run() { # label batch env...
  label=$1; bs=$2; shift; shift
  env "$@" timeout 600 python bench.py --no-fp16 --no-cpu-baseline --no-roofline --no-batch8 --no-dropin --steps 50 --batch $bs 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$label bs$bs', d['ms_per_step'])" >> $out/bench.txt
}
run off 1 MIXDQ_PREFETCH=0
run on40 1 MIXDQ_PREFETCH=1
run on20 1 MIXDQ_PREFETCH=1 MIXDQ_PREFETCH_MB=20
run on80 1 MIXDQ_PREFETCH=1 MIXDQ_PREFETCH_MB=80
run off 1 MIXDQ_PREFETCH=0
run on40 1 MIXDQ_PREFETCH=1
run off 2 MIXDQ_PREFETCH=0
run on40 2 MIXDQ_PREFETCH=1
run on40_rows8k 2 MIXDQ_PREFETCH=1 MIXDQ_PREFETCH_MAX_ROWS=8192
run off 4 MIXDQ_PREFETCH=0
run on40 4 MIXDQ_PREFETCH=1
run on40_rows16k 4 MIXDQ_PREFETCH=1 MIXDQ_PREFETCH_MAX_ROWS=16384
cat $out/pytest.txt $out/bench.txt
