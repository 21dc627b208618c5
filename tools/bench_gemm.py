#!/usr/bin/env python3
"""Micro-benchmark of the INT8 GEMM / conv kernel configurations on the dominant SDXL shapes
(SURVEY.md Appendix A).  Runs on the GPU box:  python tools/bench_gemm.py [--conv] [--bs B]

For every shape x configuration: checks the result bit-for-bit against configuration 1 and
reports the mean kernel time over a hipGraph of back-to-back launches (HIP events)."""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mixdq_amd._C as C  # noqa: E402

DEV = "cuda:0"
LINEAR = [  # (count per 1024px image, M per image, N, K)
    (372, 1024, 1280, 1280), (60, 1024, 10240, 1280), (60, 1024, 1280, 5120),
    (70, 4096, 640, 640), (10, 4096, 5120, 640), (10, 4096, 640, 2560),
    (120, 77, 1280, 2048), (20, 77, 640, 2048), (17, 1, 1280, 1280),
    (60, 1024, 3840, 1280), (10, 4096, 1920, 640), (60, 77, 2560, 2048),   # fused q|k|v, k|v
]
CONV = [  # (count, H=W, Cin, Cout, ksize, stride)
    (10, 32, 1280, 1280, 3, 1), (7, 128, 320, 320, 3, 1), (6, 64, 640, 640, 3, 1),
    (2, 32, 2560, 1280, 3, 1), (1, 64, 1280, 1280, 3, 1), (1, 128, 640, 640, 3, 1),
    (2, 128, 640, 320, 3, 1), (1, 64, 1920, 640, 3, 1), (1, 128, 960, 320, 3, 1),
    (3, 32, 1280, 1280, 1, 1),
]


def timeit(fn, iters=20):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / iters)
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--bs", type=int, default=1)
    ap.add_argument("--conv", action="store_true")
    ap.add_argument("--cfgs", default="")
    ap.add_argument("--w4", action="store_true", help="packed 4-bit weights (MIXDQ_FLAG_W4)")
    args = ap.parse_args()
    cfgs = [int(c) for c in args.cfgs.split(",")] if args.cfgs else sorted(C.IGEMM_CONFIGS)
    g = torch.Generator(device="cpu").manual_seed(0)
    zero = torch.zeros((), device=DEV)
    results = []
    shapes = CONV if args.conv else LINEAR
    for shp in shapes:
        if args.conv:
            cnt, hw, cin, cout, ks, stride = shp
            x = torch.randint(-128, 128, (args.bs, cin, hw, hw), generator=g, dtype=torch.int8
                              ).to(DEV).contiguous(memory_format=torch.channels_last)
            w = torch.randint(-128, 128, (cout, cin, ks, ks), generator=g, dtype=torch.int8
                              ).to(DEV).contiguous(memory_format=torch.channels_last)
            pad = ks // 2
            wsum = w.float().sum(dim=1, keepdim=True)
            if args.w4:
                from mixdq_amd.nn.utils import pack_w4
                w = pack_w4((w >> 4).permute(0, 2, 3, 1).contiguous()).permute(0, 3, 1, 2)
            table = C.conv_border_table(wsum) if pad else None
            b0 = None if pad else w.float().sum(dim=[1, 2, 3])
            sc = torch.rand(cout, generator=g).to(DEV) * 1e-4
            bias = torch.rand(cout, generator=g).half().to(DEV)
            ops = 2.0 * args.bs * (hw // stride) ** 2 * cout * cin * ks * ks

            def run(cfg):
                return C.qconv2d_w8_a8_ohalf(x, w, sc, zero, zero, sc, wsum if pad else None, b0,
                                             bias, stride, pad, 1, _table=table, _cfg=cfg,
                                             _w4=args.w4)
            label = f"conv {hw}x{hw} {cin}->{cout} k{ks}"
        else:
            cnt, M, N, K = shp
            M *= args.bs
            a = torch.randint(-128, 128, (M, K), generator=g, dtype=torch.int8).to(DEV)
            w = torch.randint(-128, 128, (N, K), generator=g, dtype=torch.int8).to(DEV)
            if args.w4:
                from mixdq_amd.nn.utils import pack_w4
                w = pack_w4(w >> 4)
            sc = torch.rand(N, generator=g).to(DEV) * 1e-4
            b0 = torch.rand(N, generator=g).to(DEV) * 100
            bias = torch.rand(N, generator=g).half().to(DEV)
            ops = 2.0 * M * N * K

            def run(cfg):
                return C.qlinear_w8_a8_ohalf(a, w, sc, zero, zero, b0, sc, b0, bias, _cfg=cfg,
                                             _w4=args.w4)
            label = f"lin M{M} N{N} K{K}"
        ref = run(1)
        row = dict(shape=label, count=cnt, gops=ops / 1e9, us={})
        for cfg in cfgs:
            try:
                out = run(cfg)
            except RuntimeError as e:           # a tile that does not take this problem (halo: 3x3 only)
                row["us"][cfg] = f"n/a ({str(e)[-40:]})"
                continue
            ok = torch.equal(out, ref)
            us = timeit(lambda: run(cfg))
            row["us"][cfg] = round(us, 2)
            if not ok:
                row["us"][cfg] = f"MISMATCH({us:.1f})"
        auto = timeit(lambda: run(0))
        row["auto_us"] = round(auto, 2)
        best = min((v, k) for k, v in row["us"].items() if not isinstance(v, str))
        row["best"] = f"cfg{best[1]} {best[0]}us {ops / best[0] / 1e6:.0f} TOPS"
        results.append(row)
        print(json.dumps(row), flush=True)
    tot_auto = sum(r["count"] * r["auto_us"] for r in results)
    tot_best = sum(r["count"] * min(v for v in r["us"].values() if not isinstance(v, str))
                   for r in results)
    print(f"weighted total per image: auto {tot_auto / 1e3:.2f} ms, best-per-shape {tot_best / 1e3:.2f} ms")


if __name__ == "__main__":
    main()
