#!/usr/bin/env python3
"""LayerNorm+quantize / GroupNorm+SiLU+quantize launches of the batch-8 step next to a plain
fp16 -> int8 conversion of the same tensor (same bytes read and written, no reduction): us per launch,
hipGraph chain of 20."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mixdq_amd._C as C  # noqa: E402
from tools.bench_floor import timed  # noqa: E402

DEV = "cuda:0"
one, z = torch.ones((), device=DEV), torch.zeros((), device=DEV)
for M, Cc in ((8192, 1280), (32768, 640), (1024, 1280), (4096, 640), (16384, 1280)):
    x = torch.randn(M, Cc, device=DEV).half()
    w, b = torch.ones(Cc, device=DEV).half(), torch.zeros(Cc, device=DEV).half()
    o = torch.empty(M, Cc, dtype=torch.int8, device=DEV)
    t_ln = timed(lambda: C.layernorm_quantize(x, w, b, 1e-5, [(one, z)]), 20)
    t_cp = timed(lambda: o.copy_(x), 20)
    t_q = timed(lambda: C.quantize_per_tensor_to_int8(x, one, z), 20)
    by = 3 * M * Cc
    print(f"LN ({M},{Cc}) rows={os.environ.get('MIXDQ_LN_ROWS', 'auto')}: {t_ln:6.2f} us = {by / t_ln / 1e6:5.2f} TB/s | torch f16->i8 copy {t_cp:6.2f} us | quantize {t_q:6.2f} us", flush=True)
for N, HW, Cc in ((8, 128, 320), (8, 64, 640), (8, 32, 1280), (1, 128, 320), (1, 128, 960), (1, 64, 640), (1, 64, 1920), (1, 32, 1280), (2, 128, 320)):
    x = torch.randn(N, Cc, HW, HW, device=DEV).half().contiguous(memory_format=torch.channels_last)
    w, b = torch.ones(Cc, device=DEV).half(), torch.zeros(Cc, device=DEV).half()
    o = torch.empty(N, HW, HW, Cc, dtype=torch.int8, device=DEV)
    t_gn = timed(lambda: C.groupnorm_silu_quantize(x, 32, w, b, 1e-5, one, z), 20)
    t_gn0 = timed(lambda: C.groupnorm_silu_quantize(x, 32, w, b, 1e-5, one, z, silu=False), 20)
    t_cp = timed(lambda: o.copy_(x.permute(0, 2, 3, 1)), 20)
    by = 5 * x.numel()          # stats read + apply read + int8 write
    print(f"GN ({N},{HW}x{HW},{Cc}): silu {t_gn:6.2f} us = {by / t_gn / 1e6:5.2f} TB/s | no silu {t_gn0:6.2f} | torch f16->i8 copy {t_cp:6.2f} us", flush=True)
