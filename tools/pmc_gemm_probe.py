#!/usr/bin/env python3
"""One launch shape, a few launches, for rocprofv3 --pmc passes (program directly after `--`):
    rocprofv3 --pmc <counters> --output-format csv -d out -- python3 tools/pmc_gemm_probe.py lin   M N K [cfg]
                                                              ... python3 tools/pmc_gemm_probe.py geglu M N K [cfg]
                                                              ... python3 tools/pmc_gemm_probe.py conv  IMG HW CIN COUT [cfg]
                                                              ... python3 tools/pmc_gemm_probe.py attn  B T C   (self-attention, heads = C / 64)
                                                              ... python3 tools/pmc_gemm_probe.py linattn B T C  (to_q + cross-attention, 77 keys)
                                                              ... python3 tools/pmc_gemm_probe.py ln    M N K   (GEMM + residual + LayerNorm + quantize)
                                                              ... python3 tools/pmc_gemm_probe.py f16in M N K   (quantize-in-prologue GEMM)
Inputs are made on the CPU and copied; no PyTorch GPU kernel is launched (rocprofv3 --pmc segfaults
inside some torch reduction launches on this image)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mixdq_amd._C as C  # noqa: E402

kind = sys.argv[1]
a1, a2, a3 = (int(v) for v in sys.argv[2:5])
rest = [int(v) for v in sys.argv[5:]]
g = torch.Generator().manual_seed(0)
one, zero = torch.ones(()).to("cuda"), torch.zeros(()).to("cuda")
REPS = 4
if kind in ("lin", "geglu"):
    M, N, K = a1, a2, a3
    cfg = rest[0] if rest else 0
    a = torch.randint(-128, 128, (M, K), generator=g, dtype=torch.int8).to("cuda")
    w = torch.randint(-128, 128, (N, K), generator=g, dtype=torch.int8).to("cuda")
    # GEGLU: outputs of a few units (|gate| < 8 almost everywhere, as calibrated activations are; the
    # epilogue's beyond-the-table path is timed by tools/gpu_geglu.sh); plain: range does not matter
    sc = (torch.rand(N, generator=g) * (1e-5 if kind == "geglu" else 1e-4)).to("cuda")
    for _ in range(REPS):
        if kind == "geglu":
            C.qlinear_geglu(a, w, sc, sc, None, one, zero, _cfg=cfg)
        else:
            C.qlinear_w8_a8_ohalf(a, w, sc, zero, zero, sc, sc, sc, None, _cfg=cfg)
elif kind == "linattn":
    B, T, Cc = a1, a2, a3
    x = torch.randint(-128, 128, (B, T, Cc), generator=g, dtype=torch.int8).to("cuda")
    w = torch.randint(-128, 128, (Cc, Cc), generator=g, dtype=torch.int8).to("cuda")
    sc = (torch.rand(Cc, generator=g) * 2e-5).to("cuda")
    kv = torch.randn(B, 77, 2 * Cc, generator=g).half().to("cuda")
    s_inv = torch.full((), 20.0).to("cuda")
    for _ in range(REPS):
        C.qlinear_attention(x, w, sc, sc, kv[..., :Cc], kv[..., Cc:], s_inv, zero)
elif kind == "ln":
    M, N, K = a1, a2, a3
    a = torch.randint(-128, 128, (M, K), generator=g, dtype=torch.int8).to("cuda")
    w = torch.randint(-128, 128, (N, K), generator=g, dtype=torch.int8).to("cuda")
    sc = (torch.rand(N, generator=g) * 4e-5).to("cuda")
    res = torch.randn(M, N, generator=g).half().to("cuda")
    gm, bt = torch.ones(N).half().to("cuda"), torch.zeros(N).half().to("cuda")
    ws = torch.zeros(int(C._lib.mixdq_qlinear_ln_workspace_bytes(M, N)), dtype=torch.uint8).to("cuda")
    s_inv = torch.full((), 30.0).to("cuda")
    for _ in range(REPS):
        C.qlinear_ln(a, w, sc, sc, None, res, gm, bt, 1e-5, [(s_inv, zero)], ws)
elif kind == "f16in":
    M, N, K = a1, a2, a3
    x = torch.randn(M, K, generator=g).half().to("cuda")
    w = torch.randint(-128, 128, (N, K), generator=g, dtype=torch.int8).to("cuda")
    sc = (torch.rand(N, generator=g) * 1e-4).to("cuda")
    s_inv = torch.full((), 30.0).to("cuda")
    for _ in range(REPS):
        C.qlinear_f16in(x, s_inv, zero, w, sc, sc, None)
elif kind == "attn":
    B, T, Cc = a1, a2, a3
    qkv = (torch.randn(B, T, 3 * Cc, generator=g) * 1.0).half().to("cuda")     # the fused q|k|v projection's layout
    s_inv = torch.full((), 20.0).to("cuda")
    for _ in range(REPS):
        C.attention_f16(qkv[..., :Cc], qkv[..., Cc:2 * Cc], qkv[..., 2 * Cc:], Cc // 64, s_inv, zero)
else:
    NI, HW, CIN = a1, a2, a3
    COUT, cfg = rest[0], (rest[1] if len(rest) > 1 else 0)
    x = torch.randint(-128, 128, (NI, HW, HW, CIN), generator=g, dtype=torch.int8).to("cuda").permute(0, 3, 1, 2)
    w = torch.randint(-128, 128, (COUT, 3, 3, CIN), generator=g, dtype=torch.int8).to("cuda").permute(0, 3, 1, 2)
    wsum = torch.randint(-128, 128, (COUT, 1, 3, 3), generator=g, dtype=torch.int32).float().to("cuda")
    sc = (torch.rand(COUT, generator=g) * 1e-4).to("cuda")
    table = C.conv_border_table(wsum)
    for _ in range(REPS):
        C.qconv2d_w8_a8_ohalf(x, w, sc, zero, one, sc, wsum, None, None, 1, 1, 1, _table=table, _cfg=cfg)
torch.cuda.synchronize()
