#!/bin/bash
# Round-3 experiment K: where the big GEMM+GEGLU tiles spend their time (stamps), GEGLU on the
# four-phase 256x256 tile with this round's epilogue, the 5-stage 128x320x64 tile.
out=gpurun_out/r03_k
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
python - > $out/geglu_cfgs.txt 2>&1 <<'PY'
import sys, torch
sys.path.insert(0, '.')
import mixdq_amd._C as C
from tools.bench_floor import timed
g = torch.Generator(device="cpu").manual_seed(0)
one, z = torch.ones((), device="cuda"), torch.zeros((), device="cuda")
for (M, N, K), cfgs in (((1024, 10240, 1280), (0, 25, 46, 47)), ((4096, 5120, 640), (0, 25, 46, 47, 35)),
                        ((8192, 10240, 1280), (0, 13, 18, 70, 20, 25)), ((32768, 5120, 640), (0, 13, 18, 70, 20, 25))):
    a = torch.randint(-128, 128, (M, K), generator=g, dtype=torch.int8).cuda()
    w = torch.randint(-128, 128, (N, K), generator=g, dtype=torch.int8).cuda()
    sc = torch.rand(N, generator=g).cuda() * 1e-4
    row = {}
    for cfg in cfgs:
        try:
            row[cfg] = round(timed(lambda: C.qlinear_geglu(a, w, sc, sc, None, one, z, _cfg=cfg), 50), 2)
        except RuntimeError as e:
            row[cfg] = str(e)[:30]
    plain = {}
    for cfg in (0, 13, 70):
        try:
            plain[cfg] = round(timed(lambda: C.qlinear_w8_a8_ohalf(a, w, sc, z, z, sc, sc, sc, None, _cfg=cfg), 50), 2)
        except RuntimeError as e:
            plain[cfg] = str(e)[:30]
    print((M, N, K), "geglu:", row, "| plain:", plain, flush=True)
PY
export MIXDQ_HIP_LIB=$PWD/build/stamp/libmixdq_stamp.so
for spec in "8192 10240 1280 --geglu --cfg 13" "8192 10240 1280 --geglu --cfg 70" "8192 10240 1280 --cfg 70" "8192 10240 1280 --cfg 13" \
            "1024 10240 1280 --geglu --cfg 25" "1024 10240 1280 --geglu --cfg 25 --cold" "1024 10240 1280 --geglu --cfg 47" "1024 10240 1280 --geglu --cfg 47 --cold"; do
  echo "== $spec" >> $out/stamps.txt
  timeout 300 python tools/stamp_report.py $spec >> $out/stamps.txt 2>&1
done
cat $out/geglu_cfgs.txt $out/stamps.txt
