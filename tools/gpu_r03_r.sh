#!/bin/bash
out=gpurun_out/r03_r
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
timeout 600 python -m pytest tests/test_fused_gpu.py -q -m gpu -k "geglu" -x 2>&1 | tail -2 > $out/pytest.txt
python - > $out/geglu.txt 2>&1 <<'PY'
import sys, torch
sys.path.insert(0, '.')
import mixdq_amd._C as C
from tools.bench_floor import timed
g = torch.Generator(device="cpu").manual_seed(0)
one, z = torch.ones((), device="cuda"), torch.zeros((), device="cuda")
for (M, N, K) in ((1024, 10240, 1280), (4096, 5120, 640), (8192, 10240, 1280), (32768, 5120, 640), (16384, 10240, 1280)):
    a = torch.randint(-128, 128, (M, K), generator=g, dtype=torch.int8).cuda()
    w = torch.randint(-128, 128, (N, K), generator=g, dtype=torch.int8).cuda()
    row = {}
    for sc_mag, tag in ((1e-4, "wide"), (1e-5, "typical")):
        sc = torch.rand(N, generator=g).cuda() * sc_mag
        row[tag] = round(timed(lambda: C.qlinear_geglu(a, w, sc, sc, None, one, z), 50), 2)
    plain = round(timed(lambda: C.qlinear_w8_a8_ohalf(a, w, sc, z, z, sc, sc, sc, None), 50), 2)
    print((M, N, K), row, "| plain", plain, flush=True)
PY
cat $out/pytest.txt $out/geglu.txt
