#!/bin/bash
# Round 6: swap_glue's operand hand-off (nn/glue.py OPERAND_PAIRS) and kept BOS buffers: tests, then the default bench
# line's drop-in legs (own quantize launches vs handed-on operands, same process, same box).  -> gpurun_out/r06_handoff/
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
out=gpurun_out/r06_handoff
rm -rf $out; mkdir -p $out
timeout 900 python -m pytest tests/test_glue_gpu.py tests/test_modules_gpu.py tests/test_unet_gpu.py tests/test_f16in_gpu.py -q -m gpu 2>&1 | tail -15 | tee $out/pytest.txt
for rep in 1; do
  timeout 900 python bench.py --no-cpu-baseline --no-roofline --no-batch8 --no-lnchain --steps 20 > $out/bench_$rep.json 2> $out/bench_$rep.err
  python3 - $out/bench_$rep.json <<'PY' | tee -a $out/ab.txt
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print({k: (round(v, 3) if isinstance(v, float) else v) for k, v in d.items() if (k.startswith("dropin") and "attention" not in k) or k.startswith("speedup") or k in ("ms_per_step",)})
    print("fp16", d.get("fp16"))
except Exception as e:
    print("ERR", e); print(open(sys.argv[1].replace(".json", ".err")).read()[-2000:])
PY
done
