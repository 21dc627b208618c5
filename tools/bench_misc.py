#!/usr/bin/env python3
"""Micro-benchmarks: quantize kernel over typical SDXL activation sizes; igemm K-sweep to split
fixed cost from per-K-tile cost.  GPU box only."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mixdq_amd._C as C  # noqa: E402
from tools.bench_gemm import timeit  # noqa: E402

DEV = "cuda:0"


def main():
    s_inv = torch.tensor(20.0, device=DEV)
    zp = torch.tensor(3.0, device=DEV)
    print("quantize (dense):")
    for shape in [(1024, 1280), (4096, 640), (16384, 320), (1024, 5120), (77, 2048), (1, 1280),
                  (16384, 1920), (8, 16384, 320)]:
        x = torch.randn(*shape, device=DEV).half()
        us = timeit(lambda: C.quantize_per_tensor_to_int8(x, s_inv, zp), 50)
        n = x.numel()
        print(f"  {str(shape):>20}  {us:7.2f} us   {3 * n / us / 1e3:8.1f} GB/s")
    print("empty kernel-ish floor (1 element):")
    x = torch.randn(8, device=DEV).half()
    print(f"  {timeit(lambda: C.quantize_per_tensor_to_int8(x, s_inv, zp), 50):7.2f} us")
    print("igemm K sweep, M=1024 N=1280:")
    zero = torch.zeros((), device=DEV)
    for K in (128, 256, 512, 1024, 1280, 2560, 5120):
        a = torch.randint(-128, 128, (1024, K), dtype=torch.int8, device=DEV)
        w = torch.randint(-128, 128, (1280, K), dtype=torch.int8, device=DEV)
        sc = torch.rand(1280, device=DEV)
        row = []
        for cfg in (1, 4, 5, 9):
            row.append(timeit(lambda: C.qlinear_w8_a8_ohalf(a, w, sc, zero, zero, sc, sc, sc, None,
                                                            _cfg=cfg), 50))
        print(f"  K={K:5d}  " + "  ".join(f"cfg{c}:{u:6.2f}" for c, u in zip((1, 4, 5, 9), row)))


if __name__ == "__main__":
    main()
