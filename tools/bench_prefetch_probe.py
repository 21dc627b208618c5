#!/usr/bin/env python3
"""Does touching a weight tensor one launch ahead (HBM -> Infinity Cache) make the GEMM that reads it run at
its MALL-warm speed?  Chains over 80 distinct weight tensors (HBM-cold in rotation):
  A  gemm(w[i])                                   cold
  B  touch(w[i+1]); gemm(w[i])                    the next tensor read by a trivial kernel first
  C  touch(w[i+1])                                the touch alone
B - C against A is what the GEMM gained; us per iteration."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mixdq_amd._C as C  # noqa: E402
from tools.bench_weight_residency import timed  # noqa: E402

DEV = "cuda:0"
gen = torch.Generator(device="cpu").manual_seed(0)
s, z = torch.ones((), device=DEV), torch.zeros((), device=DEV)
n = 80
for kind, M, N, K in (("geglu", 1024, 10240, 1280), ("linear", 1024, 1280, 5120), ("linear", 1024, 3840, 1280)):
    a = torch.randint(-128, 128, (M, K), generator=gen, dtype=torch.int8).to(DEV)
    sc = (torch.rand(N, generator=gen) * 1e-4).to(DEV)
    b0 = torch.zeros(N, device=DEV)
    ws = [torch.randint(-128, 128, (N, K), generator=gen, dtype=torch.int8).to(DEV) for _ in range(n)]
    sink = torch.zeros(N * K // 16, dtype=torch.int32, device=DEV)
    if kind == "geglu":
        gemm = lambda i: C.qlinear_geglu(a, ws[i % n], sc, b0, None, s, z)       # noqa: E731
    else:
        gemm = lambda i: C.qlinear_w8_a8_ohalf(a, ws[i % n], sc, z, z, b0, sc, b0, None)   # noqa: E731
    # touch: one elementwise pass that reads every byte of the tensor (strided view: a quarter of the
    # dwords, still every 128-byte line... no: every line needs every 32nd dword; read them all, simply)
    touch = lambda i: torch.bitwise_or(ws[(i + 1) % n].view(torch.int32).view(-1)[: sink.numel() * 4].view(4, -1)[0], 0, out=sink)  # noqa: E731
    touch_all = lambda i: ws[(i + 1) % n].view(torch.int32).view(-1).max()    # noqa: E731
    A = timed(gemm)
    Cq = timed(touch_all)
    B = timed(lambda i: (touch_all(i), gemm(i)))
    print(json.dumps({"launch": f"{kind} ({M},{N},{K})", "A_gemm_cold": A, "B_touch_then_gemm": B, "C_touch": Cq,
                      "gemm_after_touch": round(B - Cq, 2)}), flush=True)
    del ws
