#!/usr/bin/env python3
"""GEMM(+residual) -> LayerNorm+quantize as two launches against mixdq_qlinear_w8a8_ln (one launch), us per
layer inside a captured graph: a dependent chain of L layers on 40 weight tensors in rotation -- each layer's
INT8 operand is the previous layer's LayerNorm output, its residual the previous layer's FP16 rows (the
transformer block's pattern).

    python tools/bench_ln_gemm.py
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

SHAPES = [("to_out.0 / proj 1280", 1024, 1280, 1280), ("ff.net.2 1280", 1024, 1280, 5120),
          ("to_out.0 640", 4096, 640, 640), ("ff.net.2 640", 4096, 640, 2560), ("batch 2, 1280", 2048, 1280, 1280)]


def main():
    import torch
    import mixdq_amd._C as C
    dev = "cuda:0"
    gen = torch.Generator(device="cpu").manual_seed(0)

    def timed(fn, L=120, reps=5):
        fn(0)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for i in range(L):
                fn(i)
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best = 1e9
        for _ in range(reps):
            e0.record()
            g.replay()
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e3 / L)
        return round(best, 2)

    s, z = torch.full((), 30.0, device=dev), torch.full((), 3.0, device=dev)
    for name, M, N, K in SHAPES:
        nw = max(2, min(40, (400 << 20) // (N * K)))
        ws = [torch.randint(-128, 128, (N, K), generator=gen, dtype=torch.int8).to(dev) for _ in range(nw)]
        sc = (torch.rand(N, generator=gen) * 4e-5 + 2e-5).to(dev)
        b0 = torch.zeros(N, device=dev)
        gm, bt = torch.ones(N, device=dev).half(), torch.zeros(N, device=dev).half()
        x0 = torch.randn(M, N, generator=gen).half().to(dev)
        q0 = torch.randint(-128, 128, (M, K), generator=gen, dtype=torch.int8).to(dev)
        wsb = C.qlinear_ln_workspace(M, N, dev)
        st = {}

        def reset():
            st["x"], st["q"] = x0, q0

        def pair(i):
            y = C.qlinear_w8_a8_ohalf(st["q"], ws[i % nw], sc, z, z, sc, sc, b0, None, _residual=st["x"])
            outs, _ = C.layernorm_quantize(y, gm, bt, 1e-5, [(s, z)])
            st["x"] = y
            if K == N:
                st["q"] = outs[0]

        def one(i):
            y, outs, _ = C.qlinear_ln(st["q"], ws[i % nw], sc, b0, None, st["x"], gm, bt, 1e-5, [(s, z)], wsb)
            st["x"] = y
            if K == N:
                st["q"] = outs[0]

        def gemm(i):
            st["x"] = C.qlinear_w8_a8_ohalf(st["q"], ws[i % nw], sc, z, z, sc, sc, b0, None, _residual=st["x"])

        r = dict(shape=name, M=M, N=N, K=K)
        for key, fn in (("pair_us", pair), ("fused_us", one), ("gemm_res_only_us", gemm)):
            reset()
            r[key] = timed(fn)
        r["cfg_fused"] = int(C._lib.mixdq_qlinear_ln_select_id(M, N, K))
        r["cfg_pair"] = C.igemm_select_id(M, N, K, K)
        print(json.dumps(r), flush=True)
        del ws


if __name__ == "__main__":
    main()
