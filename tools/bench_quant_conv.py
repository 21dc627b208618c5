#!/usr/bin/env python3
"""What the quantize launch in front of a 3x3 halo conv costs (is a quantize-in-prologue form of the
halo conv worth building?): us per layer in a hipGraph over rotating weights -- the conv alone on a pre-quantized
operand, and the reference's pair quantize -> conv (nn/Conv2d.py:294-311) on the FP16 tensor.

    python tools/bench_quant_conv.py [--bs B]"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mixdq_amd._C as C  # noqa: E402
from tools.bench_floor import timed  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--bs", type=int, default=1)
a = ap.parse_args()
g = torch.Generator(device="cpu").manual_seed(0)
one, z = torch.ones((), device="cuda"), torch.zeros((), device="cuda")
s_inv = torch.full((), 20.0, device="cuda")
for hw, cin, cout in ((64, 640, 640), (32, 1280, 1280), (128, 320, 320)):
    x16 = torch.randn(a.bs, hw, hw, cin, generator=g).half().cuda().permute(0, 3, 1, 2)
    x8 = C.quantize_per_tensor_to_int8(x16, s_inv, z)
    ws = [torch.randint(-128, 128, (cout, 3, 3, cin), generator=g, dtype=torch.int8).cuda().permute(0, 3, 1, 2)
          for _ in range(max(2, (200 << 20) // (cout * cin * 9)))]
    wsum = ws[0].float().sum(dim=1, keepdim=True)
    table = C.conv_border_table(wsum)
    sc = torch.rand(cout, generator=g).cuda() * 1e-4
    i = [0]

    def conv(xq):
        i[0] += 1
        return C.qconv2d_w8_a8_ohalf(xq, ws[i[0] % len(ws)], sc, z, one, sc, wsum, None, None, 1, 1, 1, _table=table)

    alone = timed(lambda: conv(x8), 60)
    pair = timed(lambda: conv(C.quantize_per_tensor_to_int8(x16, s_inv, z)), 60)
    quant = timed(lambda: C.quantize_per_tensor_to_int8(x16, s_inv, z), 200)
    th, tw, bn = C.HALO_TILES[C.conv_halo_select(a.bs, hw, hw, cin, cout, 3, 3, 1, 1)]
    tiles_n = (cout + bn - 1) // bn
    redo = tiles_n * (th + 2) * (tw + 2) / (th * tw)          # times a pixel's channels would be re-quantized
    print(json.dumps({"conv": f"{a.bs}x{hw}x{hw}x{cin}->{cout}", "halo_tile": [th, tw, bn], "conv_alone_us": round(alone, 2),
                      "quantize_then_conv_us": round(pair, 2), "quantize_alone_us": round(quant, 2),
                      "cost_of_the_quantize_launch_us": round(pair - alone, 2),
                      "requantizations_per_element_if_fused": round(redo, 1)}), flush=True)
