#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
out=gpurun_out/r06_geglu_tab
rm -rf $out; mkdir -p $out
timeout 900 python -m pytest tests/test_fused_gpu.py tests/test_glue_gpu.py -x -q -m gpu -k "geglu or gelu or glue or swapped" > $out/pytest.txt 2>&1
tail -4 $out/pytest.txt
python3 - <<'PY' | tee $out/bench.txt
import os, sys, subprocess
code = r'''
import sys, torch
sys.path.insert(0, ".")
import mixdq_amd._C as C
from tools.bench_floor import timed
one, z = torch.ones((), device="cuda"), torch.zeros((), device="cuda")
for M, D in ((1024, 5120), (4096, 2560), (8192, 5120), (512, 5120)):
    h = torch.randn(M, 2 * D, device="cuda").half()
    print("geglu", (M, D), "fp16 out %.2f us" % timed(lambda: C.geglu_quantize(h, want_f16=True), 60),
          "| int8 out %.2f us" % timed(lambda: C.geglu_quantize(h, one, z), 60), flush=True)
'''
for flag in ("0", "1", "0", "1"):
    print("== MIXDQ_GEGLU_TAB=" + flag, flush=True)
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, MIXDQ_GEGLU_TAB=flag), stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
    print(r.stdout, end="", flush=True)
PY
for rep in 1 2; do for m in 0 -1; do
  e="MIXDQ_GEGLU_TAB=$m"; [ $m = -1 ] && e="MIXDQ_UNUSED=1"
  env $e timeout 900 python bench.py --no-fuse --swap-glue --no-fp16 --no-cpu-baseline --no-roofline --no-dropin --no-lnchain --no-batch8 --steps 40 > $out/b.json 2> $out/b.err
  python3 - $out/b.json $m $rep <<'PY' | tee -a $out/step_ab.txt
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("swap_glue drop-in step, GEGLU table", "off" if sys.argv[2] == "0" else "default", "rep", sys.argv[3], "ms %.3f" % d["ms_per_step"])
except Exception as e:
    print(sys.argv[2], "ERR", e)
PY
done; done
