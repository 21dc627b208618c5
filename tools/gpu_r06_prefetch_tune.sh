#!/bin/bash
# Round 6: the prefetch payload's budget after the attention XCD map (the attention's own L2 traffic fell 5x) -> gpurun_out/r06_pftune/
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
out=gpurun_out/r06_pftune
rm -rf $out; mkdir -p $out
run() {  # tag, batch, env assignments...
  tag=$1; bs=$2; shift 2
  env "$@" timeout 900 python bench.py --batch $bs --no-fp16 --no-cpu-baseline --no-roofline --no-dropin --no-lnchain --no-batch8 --steps 40 > $out/$tag.json 2> $out/$tag.err
  python3 - $out/$tag.json "$tag" <<'PY' | tee -a $out/ab.txt
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], "ms %.3f" % d["ms_per_step"])
except Exception as e:
    print(sys.argv[2], "ERR", e)
PY
}
for rep in 1 2; do
  run bs1_mb48_$rep 1 MIXDQ_PREFETCH_MB=48
  run bs1_mb32_$rep 1 MIXDQ_PREFETCH_MB=32
  run bs1_mb64_$rep 1 MIXDQ_PREFETCH_MB=64
  run bs1_mb96_$rep 1 MIXDQ_PREFETCH_MB=96
  run bs1_blocks128_$rep 1 MIXDQ_PREFETCH_BLOCKS=128
  run bs1_lead8_$rep 1 MIXDQ_PREFETCH_LEAD=8
  run bs8_mb48_$rep 8 MIXDQ_PREFETCH_MB=48
  run bs8_off_$rep 8 MIXDQ_PREFETCH=0
  run bs8_mb96_$rep 8 MIXDQ_PREFETCH_MB=96
done
