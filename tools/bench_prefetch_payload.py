#!/usr/bin/env python3
"""How long is the batch-1 self-attention launch with a prefetch payload of X MB?  The (1, 1024, 1024, 1280)
and (1, 4096, 4096, 640) launches in a hipGraph chain of 40, each launch with its OWN payload ranges (40 x X MB
of distinct memory, so every byte comes from HBM), us per launch.  The slope is the payload's throughput; the
knee is where it starts to outlast the attention (the planner's byte budget)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mixdq_amd._C as C  # noqa: E402
from tools.bench_weight_residency import timed  # noqa: E402

DEV = "cuda:0"
L = 40
for tq, c in ((1024, 1280), (4096, 640)):
    qkv = torch.randn(1, tq, 3 * c, device=DEV, dtype=torch.float16)
    q, k, v = qkv[..., :c], qkv[..., c:2 * c], qkv[..., 2 * c:]
    row = {"launch": f"self-attention (1, {tq}, {tq}, {c})", "us": {}}
    for mb in (0, 8, 16, 24, 32, 48, 64, 96):
        pool = [torch.empty(int(mb * 1e6), dtype=torch.int8, device=DEV).random_(0, 100) for _ in range(L)] if mb else None
        fn = lambda i: C.attention_f16(q, k, v, c // 64, _prefetch=[pool[i % L]] if pool else None)  # noqa: E731
        row["us"][f"{mb}MB"] = timed(fn, L=L)
        del pool
    print(json.dumps(row), flush=True)
