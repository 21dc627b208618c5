#!/usr/bin/env python3
"""Where do a launch's weights have to be for it to run at its warm speed?  The batch-1 GEMMs of the UNet,
each replayed in one hipGraph over n distinct weight tensors in rotation (80 launches, every tensor
re-used every n launches):

  n = 1                      the weights stay in the XCDs' L2 (what a plain micro-benchmark measures)
  n * bytes <= ~200 MB       re-used out of the 256 MB Infinity Cache (MALL), not out of L2 (32 MB)
  n * bytes  > 256 MB        from HBM every time -- the UNet: 2.6 GB of weights per step, each read once

The gap between the last two is what a weight PREFETCHER running beside the step (HBM is idle 97 % of a
batch-1 step: 2.6 GB / 11.4 ms = 0.23 TB/s) could buy; us per launch, HIP events."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mixdq_amd._C as C  # noqa: E402

DEV = "cuda:0"


def timed(fn, L=80, reps=5):
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


def main():
    gen = torch.Generator(device="cpu").manual_seed(0)
    s, z = torch.ones((), device=DEV), torch.zeros((), device=DEV)
    for kind, M, N, K in (("geglu", 1024, 10240, 1280), ("linear", 1024, 1280, 5120), ("linear", 1024, 3840, 1280),
                          ("linear", 1024, 1280, 1280)):
        a = torch.randint(-128, 128, (M, K), generator=gen, dtype=torch.int8).to(DEV)
        sc = (torch.rand(N, generator=gen) * 1e-4).to(DEV)
        b0 = torch.zeros(N, device=DEV)
        row = {"launch": f"{kind} ({M},{N},{K})", "weight_MB": round(N * K / 1e6, 1), "us": {}}
        for n in (1, 2, 4, 10, 20, 40, 80):
            if n * N * K > 1.2e9:
                continue
            ws = [torch.randint(-128, 128, (N, K), generator=gen, dtype=torch.int8).to(DEV) for _ in range(n)]
            if kind == "geglu":
                fn = lambda i: C.qlinear_geglu(a, ws[i % n], sc, b0, None, s, z)       # noqa: E731
            else:
                fn = lambda i: C.qlinear_w8_a8_ohalf(a, ws[i % n], sc, z, z, b0, sc, b0, None)   # noqa: E731
            row["us"][f"n{n}_{round(n * N * K / 1e6)}MB"] = timed(fn)
            del ws
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
