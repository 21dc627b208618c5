#!/usr/bin/env python3
"""GEMM + GEGLU + quantize launch per tile configuration (typical and wide gates) next to the plain GEMM
of the same shape, us per launch in a hipGraph chain:  python tools/bench_geglu_cfgs.py [--bs8]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mixdq_amd._C as C  # noqa: E402
from tools.bench_floor import timed  # noqa: E402

g = torch.Generator(device="cpu").manual_seed(0)
one, z = torch.ones((), device="cuda"), torch.zeros((), device="cuda")
SHAPES = (((1024, 10240, 1280), (0, 25, 27, 35)), ((4096, 5120, 640), (0, 25, 35)),
          ((2048, 10240, 1280), (0, 25, 27, 13)))
if "--bs8" in sys.argv:
    SHAPES = (((8192, 10240, 1280), (0, 70, 13)), ((32768, 5120, 640), (0, 70)))
for (M, N, K), cfgs in SHAPES:
    a = torch.randint(-128, 128, (M, K), generator=g, dtype=torch.int8).cuda()
    ws = [torch.randint(-128, 128, (N, K), generator=g, dtype=torch.int8).cuda() for _ in range(max(1, (300 << 20) // (N * K)))]
    row = {}
    for sc_mag, tag in ((1e-4, "wide"), (1e-5, "typical")):     # gates up to ~20 / up to ~2
        sc = torch.rand(N, generator=g).cuda() * sc_mag
        for cfg in cfgs:
            i = [0]

            def f():
                i[0] += 1
                return C.qlinear_geglu(a, ws[i[0] % len(ws)], sc, sc, None, one, z, _cfg=cfg)
            try:
                row[(tag, cfg)] = round(timed(f, 60), 2)
            except RuntimeError as e:
                row[(tag, cfg)] = str(e)[:30]
    print((M, N, K), f"cold weights ({len(ws)} tensors) geglu:", row, "| auto id", C.igemm_select_id(M, N, K, geglu=True), flush=True)
