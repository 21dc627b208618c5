"""Path A restatement: the reference's qdiff fake-quant simulation of one layer, on CPU.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): the FP-tolerance oracle for the HIP path and
the `cpu_baseline` leg of bench.py.  Follows
  quant_utils/qdiff/quantizer/base_quantizer.py:112-129   BaseQuantizer.forward
  quant_utils/qdiff/models/quant_layer.py:63-103          QuantLayer.forward
Pinned by tests/golden/fakequant.npz, produced by the imported reference classes themselves
(tests/golden/gen_golden.py).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


def fake_quant(x: torch.Tensor, delta: torch.Tensor, zero_point: torch.Tensor, n_bits: int,
               sym: bool) -> torch.Tensor:
    """x_int = round(x / delta) + zp; clamp; (x_q - zp) * delta  (base_quantizer.py:119-129).
    torch.round is half-to-even; the division is a true division (not a reciprocal multiply)."""
    x_int = torch.round(x / delta) + zero_point
    if sym:
        n_levels = 2 ** (n_bits - 1) - 1
        x_q = torch.clamp(x_int, -n_levels - 1, n_levels)
    else:
        x_q = torch.clamp(x_int, 0, 2 ** n_bits - 1)
    return (x_q - zero_point) * delta


def quant_layer_forward(x: torch.Tensor, weight: torch.Tensor, bias, w_delta, a_delta, a_zp,
                        w_bits: int = 8, a_bits: int = 8, conv_kwargs=None, split: int = 0,
                        w_delta_0=None, a_delta_0=None, a_zp_0=None) -> torch.Tensor:
    """QuantLayer.forward with weight_quant = act_quant = True.  The weight is fake-quantised on
    EVERY call (quant_layer.py:83-89), which is part of what the CPU baseline times.
    w_delta: per-output-channel [OC] (broadcast over the remaining dims); a_delta/a_zp scalars in
    the quantizer's own convention (zero point in [0, 2^n - 1])."""
    def wq(w, d):
        return fake_quant(w, d.reshape(-1, *([1] * (w.dim() - 1))), torch.zeros(()), w_bits, True)

    if split:
        x = torch.cat([fake_quant(x[:, :split], a_delta, a_zp, a_bits, False),
                       fake_quant(x[:, split:], a_delta_0, a_zp_0, a_bits, False)], dim=1)
        w = torch.cat([wq(weight[:, :split], w_delta), wq(weight[:, split:], w_delta_0)], dim=1)
    else:
        x = fake_quant(x, a_delta, a_zp, a_bits, False)
        w = wq(weight, w_delta)
    if conv_kwargs is None:
        return F.linear(x, w, bias)
    return F.conv2d(x, w, bias, **conv_kwargs)
