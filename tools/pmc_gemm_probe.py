#!/usr/bin/env python3
"""One GEMM shape x a few kernel configurations, a few launches each, for SQ/TCP counter passes.
    rocprofv3 --pmc <counters> --output-format csv -d out -- python3 tools/pmc_gemm_probe.py M N K cfg,cfg,..
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mixdq_amd._C as C  # noqa: E402

M, N, K = (int(v) for v in sys.argv[1:4])
cfgs = [int(c) for c in sys.argv[4].split(",")]
g = torch.Generator().manual_seed(0)
a = torch.randint(-128, 128, (M, K), generator=g, dtype=torch.int8).to("cuda")
w = torch.randint(-128, 128, (N, K), generator=g, dtype=torch.int8).to("cuda")
sc = (torch.rand(N, generator=g) * 1e-4).to("cuda")
zero = torch.zeros(()).to("cuda")
for cfg in cfgs:
    for _ in range(4):
        C.qlinear_w8_a8_ohalf(a, w, sc, zero, zero, sc, sc, sc, None, _cfg=cfg)
torch.cuda.synchronize()
