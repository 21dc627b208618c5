#!/usr/bin/env python3
"""Packed-W4 Linear (MIXDQ_FLAG_W4) under forced tile configurations on the UNet's batch-1 shapes;
hipGraph chain, us per launch; every configuration is checked against configuration 4."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mixdq_amd._C as C  # noqa: E402
from mixdq_amd.nn.utils import pack_w4  # noqa: E402
from tools.bench_floor import timed  # noqa: E402

DEV = "cuda:0"


def main():
    bs = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    cfgs = [int(c) for c in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0, 4, 6, 41, 3]
    g = torch.Generator(device="cpu").manual_seed(0)
    z = torch.zeros((), device=DEV)
    for M, N, K in ((1024 * bs, 1280, 1280), (1024 * bs, 1280, 5120), (1024 * bs, 10240, 1280),
                    (1024 * bs, 3840, 1280), (4096 * bs, 640, 640), (77 * bs, 1280, 2048)):
        a = torch.randint(-128, 128, (M, K), generator=g, dtype=torch.int8).to(DEV)
        q = torch.randint(-8, 8, (N, K), generator=g, dtype=torch.int8)
        w = pack_w4(q).to(DEV)
        sc = torch.rand(N, generator=g).to(DEV) * 1e-4
        ref = C.qlinear_w8_a8_ohalf(a, w, sc, z, z, sc, sc, sc, None, _cfg=4, _w4=True)
        row = {}
        for cfg in cfgs:
            out = C.qlinear_w8_a8_ohalf(a, w, sc, z, z, sc, sc, sc, None, _cfg=cfg, _w4=True)
            assert torch.equal(out, ref), cfg
            row[cfg] = round(timed(lambda: C.qlinear_w8_a8_ohalf(a, w, sc, z, z, sc, sc, sc, None,
                                                                 _cfg=cfg, _w4=True), 50), 2)
        print((M, N, K), row)


if __name__ == "__main__":
    main()
