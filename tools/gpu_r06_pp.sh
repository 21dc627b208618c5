#!/bin/bash
# Round 6: the persistent four-phase kernel -- tests, per-launch A/B, whole batch-8 step A/B  -> gpurun_out/r06_pp/
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
out=gpurun_out/r06_pp
rm -rf $out; mkdir -p $out
timeout 900 python -m pytest tests/test_fused_gpu.py tests/test_ops_gpu.py tests/test_glue_gpu.py -x -q -m gpu -k "persistent or 71 or cfg71 or glue or swapped" > $out/pytest.txt 2>&1
tail -12 $out/pytest.txt
timeout 600 python tools/bench_pp.py > $out/bench_pp.txt 2>&1
grep -v amdgpu.ids $out/bench_pp.txt
for rep in 1 2; do for x in 0 1; do
  MIXDQ_IGEMM_PERSIST=$x timeout 900 python bench.py --batch 8 --no-fp16 --no-cpu-baseline --no-roofline --no-dropin --no-lnchain --no-batch8 --steps 20 > $out/bench_bs8_p${x}_$rep.json 2> $out/bench_bs8_p${x}_$rep.err
  python3 - $out/bench_bs8_p${x}_$rep.json $x $rep <<'PY' | tee -a $out/step_ab.txt
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("persist", sys.argv[2], "rep", sys.argv[3], "batch 8 ms %.3f" % d["ms_per_step"])
except Exception as e:
    print(sys.argv[2], "ERR", e)
PY
done; done
