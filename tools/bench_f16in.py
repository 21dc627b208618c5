#!/usr/bin/env python3
"""quantize -> GEMM (the reference's two launches per layer) against the quantize-in-prologue GEMM
(mixdq_qlinear_f16in_w8a8), microseconds per LAYER inside a captured graph, for the UNet's Linear shapes.

Each chain is L layers in one hipGraph on 40 distinct weight tensors in rotation (cold weights, as in the
UNet), every layer reading the same FP16 activation tensor; `--chain` makes layer i's input the previous
layer's output (square shapes only: the hand-off of a tensor another kernel just wrote is part of the cost).

    python tools/bench_f16in.py [--bs 1|8] [--chain] [--cfg ID]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

SHAPES = [  # (name, M per image, N, K)
    ("attn to_q/k/v/out 1280", 1024, 1280, 1280), ("ff.net.2 1280", 1024, 1280, 5120),
    ("ff.net.0.proj 1280", 1024, 10240, 1280), ("attn 640", 4096, 640, 640),
    ("ff.net.2 640", 4096, 640, 2560), ("ff.net.0.proj 640", 4096, 5120, 640),
    ("attn2.to_k/v", 76, 1280, 2048), ("shortcut 1x1 1280<-640", 1024, 1280, 640),
    ("shortcut 1x1 320<-640", 16384, 320, 640),
    # (round 6: the other halves of the up-blocks' split shortcuts)
    ("shortcut 1x1 640<-320", 4096, 640, 320), ("shortcut 1x1 640<-1280", 4096, 640, 1280),
    ("shortcut 1x1 320<-320", 16384, 320, 320),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--bs", type=int, default=1)
    ap.add_argument("--chain", action="store_true")
    ap.add_argument("--cfg", type=int, default=0)
    ap.add_argument("--L", type=int, default=200)
    args = ap.parse_args()
    import torch
    import mixdq_amd._C as C
    dev = "cuda:0"
    gen = torch.Generator(device="cpu").manual_seed(0)

    def timed(fn, L, reps=5):
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

    s, z = torch.full((), 20.0, device=dev), torch.full((), 3.0, device=dev)
    rows = []
    for name, m1, N, K in SHAPES:
        M = m1 * args.bs
        nw = max(2, min(40, (400 << 20) // (N * K)))
        ws = [torch.randint(-128, 128, (N, K), generator=gen, dtype=torch.int8).to(dev) for _ in range(nw)]
        sc = (torch.rand(N, generator=gen) * 1e-4 + 1e-5).to(dev)
        b0 = torch.zeros(N, device=dev)
        x = torch.randn(M, K, generator=gen).half().to(dev)
        chain = args.chain and N == K
        state = {"x": x}

        def pair(i):
            q = C.quantize_per_tensor_to_int8(state["x"], s, z)
            y = C.qlinear_w8_a8_ohalf(q, ws[i % nw], sc, z, z, sc, sc, b0, None)
            if chain:
                state["x"] = y

        def one(i):
            y = C.qlinear_f16in(state["x"], s, z, ws[i % nw], sc, b0, None, _cfg=args.cfg)
            if chain:
                state["x"] = y

        def gemm_only(i, q=C.quantize_per_tensor_to_int8(x, s, z)):
            C.qlinear_w8_a8_ohalf(q, ws[i % nw], sc, z, z, sc, sc, b0, None)

        L = args.L if M * N * K < (1 << 34) else max(20, args.L // 4)
        state["x"] = x
        t_pair = timed(pair, L)
        state["x"] = x
        t_one = timed(one, L)
        t_gemm = timed(gemm_only, L)
        r = dict(shape=name, M=M, N=N, K=K, chain=chain, pair_us=t_pair, f16in_us=t_one, int8_gemm_us=t_gemm,
                 cfg_pair=C.igemm_select_id(M, N, K, K),
                 cfg_f16in=int(C._lib.mixdq_qlinear_f16in_select_id(M, N, K, 0)))
        rows.append(r)
        print(json.dumps(r), flush=True)
        del ws


if __name__ == "__main__":
    main()
