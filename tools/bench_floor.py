#!/usr/bin/env python3
"""Per-kernel floor inside a hipGraph on this chip: dependent chains of (a) a 1-element PyTorch
add, (b) the quantize kernel on 8 elements, (c) LayerNorm+quantize on [1024, 1280], (d) the W8A8
Linear at M = 1 and M = 1024 (N = K = 1280).  us per launch over 200 launches."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mixdq_amd._C as C  # noqa: E402

DEV = "cuda:0"


def timed(fn, L=200):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(L):
            fn()
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (5 * L)


def main():
    one = torch.zeros(1, device=DEV)
    x8 = torch.zeros(8, device=DEV, dtype=torch.float16)
    s = torch.ones((), device=DEV)
    z = torch.zeros((), device=DEV)
    print("torch add, 1 element      ", round(timed(lambda: one.add_(1)), 2))
    print("quantize, 8 elements      ", round(timed(lambda: C.quantize_per_tensor_to_int8(x8, s, z)), 2))
    x = torch.randn(1024, 1280, device=DEV, dtype=torch.float16)
    w = torch.ones(1280, device=DEV, dtype=torch.float16)
    print("layernorm+quantize 1024x1280", round(timed(lambda: C.layernorm_quantize(x, w, w, 1e-5, [(s, z)])), 2))
    g = torch.Generator(device="cpu").manual_seed(0)
    wt = torch.randint(-128, 128, (1280, 1280), generator=g, dtype=torch.int8).to(DEV)
    sc = torch.rand(1280, generator=g).to(DEV) * 1e-4
    for M in (1, 64, 1024):
        a = torch.randint(-128, 128, (M, 1280), generator=g, dtype=torch.int8).to(DEV)
        print(f"qlinear M={M} N=K=1280      ", round(timed(
            lambda: C.qlinear_w8_a8_ohalf(a, wt, sc, z, z, sc, sc, sc, None)), 2))
        res = torch.randn(M, 1280, device=DEV, dtype=torch.float16)
        print(f"  + residual epilogue       ", round(timed(
            lambda: C.qlinear_w8_a8_ohalf(a, wt, sc, z, z, sc, sc, sc, None, _residual=res)), 2))


if __name__ == "__main__":
    main()
