#!/usr/bin/env python3
"""The four-phase 256x256 tile, one workgroup per tile (configuration 70) against its persistent form (71,
csrc/igemm_pp.h), us per launch in a hipGraph chain over rotating weight tensors:  python tools/bench_pp.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mixdq_amd._C as C  # noqa: E402
from tools.bench_floor import timed  # noqa: E402

g = torch.Generator(device="cpu").manual_seed(0)
one, z = torch.ones((), device="cuda"), torch.zeros((), device="cuda")
GEGLU = ((8192, 10240, 1280), (32768, 5120, 640), (16384, 10240, 1280))
PLAIN = ((8192, 3840, 1280), (32768, 1920, 640), (8192, 1280, 5120), (16384, 3840, 1280))
for kind, shapes in (("geglu", GEGLU), ("plain", PLAIN)):
    for (M, N, K) in shapes:
        a = torch.randint(-128, 128, (M, K), generator=g, dtype=torch.int8).cuda()
        ws = [torch.randint(-128, 128, (N, K), generator=g, dtype=torch.int8).cuda()
              for _ in range(max(1, (300 << 20) // (N * K)))]
        sc = torch.rand(N, generator=g).cuda() * 1e-5
        row = {}
        for cfg in (70, 71, 70, 71):
            i = [0]

            def f():
                i[0] += 1
                w = ws[i[0] % len(ws)]
                if kind == "geglu":
                    return C.qlinear_geglu(a, w, sc, sc, None, one, z, _cfg=cfg)
                return C.qlinear_w8_a8_ohalf(a, w, sc, z, z, sc, sc, sc, None, _cfg=cfg)
            row.setdefault(cfg, []).append(round(timed(f, 40), 2))
        auto = C.igemm_select_id(M, N, K, geglu=kind == "geglu")
        print(kind, (M, N, K), "us per launch:", row, "| auto id", auto, "| Pop/s at best 71:",
              round(2.0 * M * N * K / min(row[71]) / 1e9, 3), flush=True)
