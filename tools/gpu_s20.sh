#!/bin/bash
# session 20: weight prefetch from the self-attention launch -- tests, batch-1 step A/B (MIXDQ_PREFETCH=0/1, NT, blocks)
out=gpurun_out/s20
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
( time timeout 1500 python -m pytest tests/test_attention_gpu.py tests/test_unet_gpu.py tests/test_unet_full_gpu.py -q -m gpu 2>&1 | tail -8 ) > $out/pytest.txt 2>&1
run() { # label env...
  label=$1; shift
  env "$@" timeout 600 python bench.py --no-fp16 --no-cpu-baseline --no-roofline --no-batch8 --no-dropin --steps 50 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$label bs1', d['ms_per_step'])" >> $out/bench.txt
}
run off MIXDQ_PREFETCH=0
run on MIXDQ_PREFETCH=1
run off MIXDQ_PREFETCH=0
run on MIXDQ_PREFETCH=1
run on_nt MIXDQ_PREFETCH=1 MIXDQ_PREFETCH_NT=1
run on_b128 MIXDQ_PREFETCH=1 MIXDQ_PREFETCH_BLOCKS=128
run on_b512 MIXDQ_PREFETCH=1 MIXDQ_PREFETCH_BLOCKS=512
run off MIXDQ_PREFETCH=0
run on MIXDQ_PREFETCH=1
cat $out/pytest.txt $out/bench.txt
