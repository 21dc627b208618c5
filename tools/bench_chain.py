#!/usr/bin/env python3
"""What does a producer -> GEMM pair cost inside the batch-1 chain, and what do cold weights cost?

A hipGraph of L dependent pairs  LayerNorm+quantize(x) -> INT8 GEMM(W_i) (+ residual x)  as the
transformer blocks run them, per shape:
   warm      every pair uses the same W (weights stay in the caches)
   cold      L distinct W (every launch streams its weights from HBM, as in the UNet)
   gemm-only the GEMMs alone on a fixed INT8 operand (cold weights): pair - this = the LN launch
   ln-only   the LayerNorm launches alone
   empty     a chain of L trivial launches (the boundary)

    python tools/bench_chain.py [--L 120]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mixdq_amd._C as C  # noqa: E402

DEV = "cuda:0"


def time_graph(build, reps=5):
    build()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        build()
    for _ in range(2):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(reps):
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3)
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--L", type=int, default=120)
    ap.add_argument("--shapes", default="1024x1280x1280,1024x3840x1280,1024x10240x1280,4096x640x640")
    args = ap.parse_args()
    g = torch.Generator(device="cpu").manual_seed(0)
    zero = torch.zeros((), device=DEV)
    s_inv, zp = torch.tensor(20.0, device=DEV), torch.tensor(3.0, device=DEV)
    flush = torch.empty(512 << 20, dtype=torch.uint8, device=DEV)
    for shp in args.shapes.split(","):
        M, N, K = (int(v) for v in shp.split("x"))
        # enough distinct weights to overflow the 256 MiB Infinity Cache between two uses
        L = min(480, max(args.L, -(-(640 << 20) // (N * K))))
        x = torch.randn(M, K, generator=g).half().to(DEV)
        gm, bt = torch.ones(K).half().to(DEV), torch.zeros(K).half().to(DEV)
        ws = [torch.randint(-128, 128, (N, K), generator=g, dtype=torch.int8).to(DEV) for _ in range(L)]
        sc = (torch.rand(N, generator=g) * 1e-4).to(DEV)
        b0 = (torch.rand(N, generator=g) * 100).to(DEV)
        res = torch.randn(M, N, generator=g).half().to(DEV) if N == K else None
        xq_fixed = torch.randint(-128, 128, (M, K), generator=g, dtype=torch.int8).to(DEV)

        def pairs(mode):
            def build():
                for i in range(L):
                    w = ws[0] if mode == "warm" else ws[i]
                    (q,), _ = C.layernorm_quantize(x, gm, bt, 1e-5, [(s_inv, zp)])
                    C.qlinear_w8_a8_ohalf(q, w, sc, zero, zero, b0, sc, b0, None, _residual=res)
            return build

        def gemm_only():
            for i in range(L):
                C.qlinear_w8_a8_ohalf(xq_fixed, ws[i], sc, zero, zero, b0, sc, b0, None, _residual=res)

        def ln_only():
            for i in range(L):
                C.layernorm_quantize(x, gm, bt, 1e-5, [(s_inv, zp)])

        row = dict(shape=shp, weights_mb=round(N * K / 2 ** 20, 2), L=L)
        for mode in ("warm", "cold"):
            flush.zero_()
            row[mode] = round(time_graph(pairs(mode)) / L, 2)
        row["gemm-only"] = round(time_graph(gemm_only) / L, 2)
        row["ln-only"] = round(time_graph(ln_only) / L, 2)
        print(json.dumps(row), flush=True)
        del ws
    t = torch.zeros(64, device=DEV)

    def empty():
        for _ in range(200):
            t.add_(1.0)
    print(json.dumps({"empty_chain_us_per_launch": round(time_graph(empty) / 200, 2)}))


if __name__ == "__main__":
    main()
