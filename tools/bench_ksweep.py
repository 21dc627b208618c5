#!/usr/bin/env python3
"""Latency anatomy of the small-tile W8A8 Linear: time vs K at fixed (M, N) in a hipGraph chain.
The slope is the cost of one K-tile, the intercept the launch + prologue + epilogue."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mixdq_amd._C as C  # noqa: E402
from tools.bench_floor import timed  # noqa: E402

DEV = "cuda:0"


def main():
    cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    g = torch.Generator(device="cpu").manual_seed(0)
    z = torch.zeros((), device=DEV)
    for M, N in ((1, 1280), (1024, 1280), (1024, 1024), (2048, 1280)):
        row = []
        for K in (128, 256, 640, 1280, 2560, 5120):
            a = torch.randint(-128, 128, (M, K), generator=g, dtype=torch.int8).to(DEV)
            w = torch.randint(-128, 128, (N, K), generator=g, dtype=torch.int8).to(DEV)
            sc = torch.rand(N, generator=g).to(DEV) * 1e-4
            row.append((K, round(timed(lambda: C.qlinear_w8_a8_ohalf(a, w, sc, z, z, sc, sc, sc, None,
                                                                     _cfg=cfg)), 2)))
        print(f"cfg {cfg} M={M} N={N}:", row)


if __name__ == "__main__":
    main()
