#!/usr/bin/env python3
"""Time vs K for the wide GEMM (M, N) = (1024, 10240) and its narrow sibling (1024, 1280) under a
few tile configurations: slope = one K-tile, intercept = launch + prologue + epilogue."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mixdq_amd._C as C  # noqa: E402
from tools.bench_floor import timed  # noqa: E402

DEV = "cuda:0"


def main():
    cfgs = [int(c) for c in sys.argv[1].split(",")] if len(sys.argv) > 1 else [25, 46, 41, 35]
    g = torch.Generator(device="cpu").manual_seed(0)
    z = torch.zeros((), device=DEV)
    for M, N in ((1024, 10240), (1024, 1280), (1024, 2560), (1024, 5120)):
        for cfg in cfgs:
            row = []
            for K in (128, 640, 1280, 2560, 5120):
                a = torch.randint(-128, 128, (M, K), generator=g, dtype=torch.int8).to(DEV)
                w = torch.randint(-128, 128, (N, K), generator=g, dtype=torch.int8).to(DEV)
                sc = torch.rand(N, generator=g).to(DEV) * 1e-4
                row.append((K, round(timed(lambda: C.qlinear_w8_a8_ohalf(
                    a, w, sc, z, z, sc, sc, sc, None, _cfg=cfg), 50), 2)))
            print(f"M={M} N={N} cfg {cfg}:", row, flush=True)


if __name__ == "__main__":
    main()
