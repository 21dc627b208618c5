#!/bin/bash
# Round 6: deeper-pipeline tiles IN THE STEP (MIXDQ_IGEMM_TUNE, no rebuild): is the 2-stage 128x320 tile's
# stop-and-wait K loop what the cold in-step launches lose to?  -> gpurun_out/r06_tune/
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
out=gpurun_out/r06_tune
rm -rf $out; mkdir -p $out
run() {  # tag, batch, tune string
  MIXDQ_IGEMM_TUNE="$3" timeout 900 python bench.py --batch $2 --no-fp16 --no-cpu-baseline --no-roofline --no-dropin --no-lnchain --no-batch8 --steps 20 > $out/$1.json 2> $out/$1.err
  python3 - $out/$1.json "$1" <<'PY' | tee -a $out/ab.txt
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], "ms %.3f" % d["ms_per_step"])
except Exception as e:
    print(sys.argv[2], "ERR", e)
PY
}
for rep in 1 2; do
  run bs8_default_$rep 8 ""
  run bs8_k5120_cfg46_$rep 8 "8192x1280x5120=46"
  run bs8_k5120_cfg47_$rep 8 "8192x1280x5120=47"
  run bs8_k1280_cfg46_$rep 8 "8192x1280x1280=46"
  run bs8_k5120_cfg71_$rep 8 "8192x1280x5120=71"
  run bs1_default_$rep 1 ""
  run bs1_geglu_cfg46_$rep 1 "1024x10240x1280=46"
  run bs1_geglu_cfg47_$rep 1 "1024x10240x1280=47"
done
