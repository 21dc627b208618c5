#!/bin/bash
# Round 6: the static tensors of the converted network in ONE allocation (mixdq_amd/arena.py, MIXDQ_WEIGHT_ARENA=1)
# against PyTorch's allocator placement, alternating processes on one box.  -> gpurun_out/r06_arena/
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
out=gpurun_out/r06_arena
rm -rf $out; mkdir -p $out
run() {  # tag, batch, arena
  MIXDQ_WEIGHT_ARENA=$3 timeout 900 python bench.py --batch $2 --no-fp16 --no-cpu-baseline --no-roofline --no-dropin --no-lnchain --no-batch8 --steps 50 > $out/$1.json 2> $out/$1.err
  python3 - $out/$1.json "$1" <<'PY' | tee -a $out/ab.txt
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], "ms %.3f" % d["ms_per_step"], "static MB", d.get("memory", {}).get("w8a8", {}).get("static_mb"))
except Exception as e:
    print(sys.argv[2], "ERR", e)
PY
}
timeout 600 python -m pytest tests/test_unet_path_a_gpu.py -x -q -m gpu 2>&1 | tail -2 | tee -a $out/ab.txt
MIXDQ_WEIGHT_ARENA=1 timeout 600 python -m pytest tests/test_unet_path_a_gpu.py tests/test_unet_gpu.py -x -q -m gpu 2>&1 | tail -2 | tee -a $out/ab.txt
for rep in 1 2 3; do
  run bs1_alloc_$rep 1 0
  run bs1_arena_$rep 1 1
done
run bs8_alloc_1 8 0
run bs8_arena_1 8 1
