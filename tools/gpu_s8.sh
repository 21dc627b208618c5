#!/bin/bash
# session 8: residual requests interleaved with the accumulator pass, streaming LayerNorm (A/B vs HEAD = build/ab_r04b)
out=gpurun_out/s8
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
B=$PWD/build/ab_r04b/libmixdq_hip.so
( time timeout 1800 python -m pytest tests/test_ops_gpu.py tests/test_fused_gpu.py tests/test_large_gpu.py tests/test_modules_gpu.py tests/test_unet_gpu.py tests/test_unet_full_gpu.py -q -m gpu -k "not over_4_gib and not shard_size" 2>&1 | tail -6 ) > $out/pytest.txt 2>&1
for v in B C B C; do
  lib=""; [ $v = B ] && lib=$B
  MIXDQ_HIP_LIB=$lib timeout 600 python bench.py --no-fp16 --no-cpu-baseline --no-roofline --steps 20 --batch 8 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v bs8', d['ms_per_step'])" >> $out/bench.txt
done
MIXDQ_LN_STREAM=0 timeout 600 python bench.py --no-fp16 --no-cpu-baseline --no-roofline --steps 20 --batch 8 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C-nostream bs8', d['ms_per_step'])" >> $out/bench.txt
MIXDQ_LN_STREAM=2 timeout 600 python bench.py --no-fp16 --no-cpu-baseline --no-roofline --steps 20 --batch 8 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C-stream2 bs8', d['ms_per_step'])" >> $out/bench.txt
MIXDQ_LN_STREAM=8 timeout 600 python bench.py --no-fp16 --no-cpu-baseline --no-roofline --steps 20 --batch 8 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C-stream8 bs8', d['ms_per_step'])" >> $out/bench.txt
for v in B C; do
  lib=""; [ $v = B ] && lib=$B
  MIXDQ_HIP_LIB=$lib timeout 600 python bench.py --no-fp16 --no-cpu-baseline --no-roofline --no-batch8 --no-dropin --steps 50 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v bs1', d['ms_per_step'])" >> $out/bench.txt
done
MIXDQ_LN_STREAM=2 timeout 600 python bench.py --no-fp16 --no-cpu-baseline --no-roofline --no-batch8 --no-dropin --steps 50 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C-stream2 bs1', d['ms_per_step'])" >> $out/bench.txt
python - > $out/ln.txt 2>&1 <<'PY'
import sys, os, torch
sys.path.insert(0, '.')
import mixdq_amd._C as C
from tools.bench_floor import timed
s, z = torch.ones((), device="cuda"), torch.zeros((), device="cuda")
for M, Cc in ((8192, 1280), (32768, 640), (4096, 640), (16384, 1280)):
    x = torch.randn(M, Cc, device="cuda", dtype=torch.float16); g = torch.ones(Cc, device="cuda", dtype=torch.float16)
    print(M, Cc, "ln us", round(timed(lambda: C.layernorm_quantize(x, g, g, 1e-5, [(s, z)]), 100), 2), "MB", (3 * M * Cc) / 1e6, flush=True)
PY
MIXDQ_LN_STREAM=0 python - >> $out/ln.txt 2>&1 <<'PY'
import sys, os, torch
sys.path.insert(0, '.')
import mixdq_amd._C as C
from tools.bench_floor import timed
s, z = torch.ones((), device="cuda"), torch.zeros((), device="cuda")
for M, Cc in ((8192, 1280), (32768, 640), (4096, 640), (16384, 1280)):
    x = torch.randn(M, Cc, device="cuda", dtype=torch.float16); g = torch.ones(Cc, device="cuda", dtype=torch.float16)
    print("nostream", M, Cc, "ln us", round(timed(lambda: C.layernorm_quantize(x, g, g, 1e-5, [(s, z)]), 100), 2), flush=True)
PY
export MIXDQ_HIP_LIB=$PWD/build/stamp/libmixdq_stamp.so
for spec in "8192 1280 1280 --cfg 27 --res" "8192 1280 1280 --cfg 25 --res"; do
  echo "== $spec" >> $out/stamps.txt
  timeout 300 python tools/stamp_report.py $spec 2>&1 | grep -v amdgpu.ids | tail -2 >> $out/stamps.txt
done
unset MIXDQ_HIP_LIB
cat $out/pytest.txt $out/bench.txt $out/ln.txt $out/stamps.txt
