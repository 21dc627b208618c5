#!/usr/bin/env python3
"""ff.net.0.proj + GEGLU + quantize in one launch ((M, 2D = 10240, K = 1280) at M = 1024 per image)
under forced tile configurations; hipGraph chain, us per launch."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mixdq_amd._C as C  # noqa: E402
from tools.bench_floor import timed  # noqa: E402

DEV = "cuda:0"


def main():
    bs = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    g = torch.Generator(device="cpu").manual_seed(0)
    one, z = torch.ones((), device=DEV), torch.zeros((), device=DEV)
    for M, N, K in ((1024 * bs, 10240, 1280), (4096 * bs, 5120, 640)):
        a = torch.randint(-128, 128, (M, K), generator=g, dtype=torch.int8).to(DEV)
        w = torch.randint(-128, 128, (N, K), generator=g, dtype=torch.int8).to(DEV)
        sc = torch.rand(N, generator=g).to(DEV) * 1e-4
        row = {}
        for cfg in (0, 25, 46, 18, 13, 35):
            try:
                row[cfg] = round(timed(lambda: C.qlinear_geglu(a, w, sc, sc, None, one, z, _cfg=cfg), 50), 2)
            except RuntimeError as e:
                row[cfg] = str(e)[:20]
        plain = round(timed(lambda: C.qlinear_w8_a8_ohalf(a, w, sc, z, z, sc, sc, sc, None), 50), 2)
        print((M, N, K), "geglu-fused by cfg:", row, "| plain GEMM (auto):", plain)


if __name__ == "__main__":
    main()
