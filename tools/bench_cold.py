#!/usr/bin/env python3
"""How much of an in-context small GEMM's time is cold weights?  Times the (1024, 1280, 1280)
W8A8 Linear over a hipGraph of 200 back-to-back launches that (a) reuse one weight tensor (warm in
L2 / Infinity Cache), (b) walk 200 distinct weight tensors (328 MB: every launch streams its weights
from HBM, as in the UNet), (c) as (b) with a prefetch kernel touching the NEXT weight on a side
stream."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mixdq_amd._C as C  # noqa: E402

DEV = "cuda:0"


def main():
    M, N, K, L = 1024, 1280, 1280, 200
    g = torch.Generator(device="cpu").manual_seed(0)
    a = torch.randint(-128, 128, (M, K), generator=g, dtype=torch.int8).to(DEV)
    ws = [torch.randint(-128, 128, (N, K), generator=g, dtype=torch.int8).to(DEV) for _ in range(L)]
    sc = torch.rand(N, generator=g).to(DEV) * 1e-4
    b0 = torch.rand(N, generator=g).to(DEV)
    zero = torch.zeros((), device=DEV)
    side = torch.cuda.Stream()

    cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 0

    def gemm(w):
        return C.qlinear_w8_a8_ohalf(a, w, sc, zero, zero, b0, sc, b0, None, _cfg=cfg)

    def build(mode):
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            main = torch.cuda.current_stream()
            for i in range(L):
                w = ws[0] if mode == "warm" else ws[i]
                gemm(w)
        return gr

    for mode in ("warm", "cold"):
        gemm(ws[0]); ws[0].view(torch.int32).sum()
        torch.cuda.synchronize()
        gr = build(mode)
        for _ in range(3):
            gr.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            gr.replay()
        e1.record()
        torch.cuda.synchronize()
        print("cfg", cfg, mode, round(e0.elapsed_time(e1) * 1e3 / (5 * L), 2), "us per GEMM")


if __name__ == "__main__":
    main()
