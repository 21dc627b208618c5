#!/bin/bash
# GEMM+GEGLU+quantize launches: parity (both epilogue variants), time per tile configuration with typical and wide gates, stamps.
out=gpurun_out/geglu
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
timeout 900 python -m pytest tests/test_fused_gpu.py tests/test_large_gpu.py -q -m gpu -k "geglu" -x 2>&1 | tail -8 > $out/pytest.txt
MIXDQ_EPILOGUE_VARIANT=B timeout 900 python -m pytest tests/test_fused_gpu.py -q -m gpu -k "geglu" -x 2>&1 | tail -4 >> $out/pytest.txt
timeout 1200 python -m pytest tests/test_unet_gpu.py tests/test_unet_full_gpu.py tests/test_modules_gpu.py -q -m gpu -x 2>&1 | tail -4 >> $out/pytest.txt
python - > $out/geglu_cfgs.txt 2>&1 <<'PY'
import sys, torch
sys.path.insert(0, '.')
import mixdq_amd._C as C
from tools.bench_floor import timed
g = torch.Generator(device="cpu").manual_seed(0)
one, z = torch.ones((), device="cuda"), torch.zeros((), device="cuda")
for (M, N, K), cfgs in (((1024, 10240, 1280), (0, 25, 46, 47, 35)), ((4096, 5120, 640), (0, 25, 46, 35, 13)), ((2048, 10240, 1280), (0, 13, 25, 35)),
                        ((8192, 10240, 1280), (0, 13, 70, 20)), ((32768, 5120, 640), (0, 13, 70))):
    a = torch.randint(-128, 128, (M, K), generator=g, dtype=torch.int8).cuda()
    w = torch.randint(-128, 128, (N, K), generator=g, dtype=torch.int8).cuda()
    row = {}
    for sc_mag, tag in ((1e-4, "far"), (1e-5, "near")):     # gates up to ~20 / up to ~2
        sc = torch.rand(N, generator=g).cuda() * sc_mag
        for cfg in cfgs:
            try:
                row[(tag, cfg)] = round(timed(lambda: C.qlinear_geglu(a, w, sc, sc, None, one, z, _cfg=cfg), 50), 2)
            except RuntimeError as e:
                row[(tag, cfg)] = str(e)[:30]
    plain = round(timed(lambda: C.qlinear_w8_a8_ohalf(a, w, sc, z, z, sc, sc, sc, None), 50), 2)
    print((M, N, K), "geglu:", row, "| plain auto:", plain, "| auto id", C.igemm_select_id(M, N, K, geglu=True), flush=True)
PY
export MIXDQ_HIP_LIB=$PWD/build/stamp/libmixdq_stamp.so
for spec in "1024 10240 1280 --geglu --cfg 25" "8192 10240 1280 --geglu --cfg 70"; do
  echo "== $spec" >> $out/stamps.txt
  timeout 300 python tools/stamp_report.py $spec 2>&1 | grep -v amdgpu.ids | tail -2 >> $out/stamps.txt
done
cat $out/pytest.txt $out/geglu_cfgs.txt $out/stamps.txt
