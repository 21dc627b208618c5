"""Min-max calibration that produces a checkpoint in the kernel format ("new_ckpt.pth").

The real calibrated checkpoint of the reference is absent (kernels/output/new_ckpt.pth is a
missing large blob) and there are no SDXL weights here, so benchmarks calibrate the synthetic
UNet themselves.  The statistics follow the reference's PTQ initialisation:

  weights      symmetric per output channel: delta = absmax / (2^(n-1) - 1), zero point 0
               (base_quantizer.py:173-176,182-183 with sdxl_turbo.yaml:17-24)
  activations  asymmetric per tensor: x_min <- min(x, 0), x_max <- max(x, 0),
               delta = (x_max - x_min) / (2^n - 1), zero_point = round(-x_min / delta) in the
               uint8 convention; running min/max with momentum 0.95 over calibration batches
               (base_quantizer.py:154-171,177-185 with sdxl_turbo.yaml:25-33)
  both         for the bit-widths [2, 4, 8] stacked on dim 0 (sdxl_turbo.yaml:7), then cast to
               fp16 and reshaped to [3, OC] / [3] (convert_ckpt.py:26-40).

Output schema:
  { "<layer>.weight_quantizer": {"delta_list": f16[3,OC], "zero_point_list": f16[3,OC]},
    "<layer>.act_quantizer":    {"delta_list": f16[3],    "zero_point_list": f16[3]},
    "<layer>.weight_quantizer_0" / ".act_quantizer_0" for split up-block conv_shortcuts }
"""
from __future__ import annotations

from collections import OrderedDict

import torch
import torch.nn as nn

MIXED_PRECISION = (2, 4, 8)
EPS = 1e-6
MOMENTUM = 0.95


def weight_quantizer_entry(weight: torch.Tensor) -> dict:
    delta_list = torch.stack([weight_delta(weight, b) for b in MIXED_PRECISION], dim=0)
    return dict(delta_list=delta_list.half(), zero_point_list=torch.zeros_like(delta_list).half())


class ActRange:
    """Running min/max of one activation tensor, with the reference's update order: while a
    quantizer is un-initialised every forward calls init_quant_params once PER bit-width of
    mixed_precision (base_quantizer.py:101-111), and each call applies the momentum update
    (base_quantizer.py:160-171).  So the statistic behind index i of delta_list has seen i more
    updates with the current batch than index 0 -- reproduced here, including its rounding."""

    def __init__(self):
        self.lo = None
        self.hi = None
        self.snap = [None] * len(MIXED_PRECISION)

    def update(self, x: torch.Tensor):
        x = x.float()
        lo, hi = x.min().clamp(max=0), x.max().clamp(min=0)
        for i in range(len(MIXED_PRECISION)):
            if self.lo is None:
                self.lo, self.hi = lo, hi
            else:
                self.lo = self.lo * MOMENTUM + lo * (1 - MOMENTUM)
                self.hi = self.hi * MOMENTUM + hi * (1 - MOMENTUM)
            self.snap[i] = (self.lo, self.hi)

    def params(self, i: int):
        """fp32 (delta, zero_point) for bit-width index i, quantizer convention (zp >= 0)."""
        lo, hi = self.snap[i]
        bits = MIXED_PRECISION[i]
        delta = (hi - lo) / (2 ** bits - 1)
        if delta < EPS:
            delta = torch.full_like(delta, EPS)
        return delta, torch.round(-lo / delta)

    def entry(self) -> dict:
        ps = [self.params(i) for i in range(len(MIXED_PRECISION))]
        return dict(delta_list=torch.stack([p[0] for p in ps]).half().reshape(3),
                    zero_point_list=torch.stack([p[1] for p in ps]).half().reshape(3))


def weight_delta(weight: torch.Tensor, bits: int) -> torch.Tensor:
    """fp32 per-output-channel step of the symmetric weight quantizer."""
    w = weight.detach().float().reshape(weight.shape[0], -1)
    absmax = torch.maximum(w.min(dim=1)[0].clamp(max=0).abs(), w.max(dim=1)[0].clamp(min=0).abs())
    delta = absmax / (2 ** (bits - 1) - 1)
    if delta.min() < EPS:        # base_quantizer.py:177-180 fills the whole tensor
        delta = torch.full_like(delta, EPS)
    return delta


def _is_bos_layer(name):
    return "attn2" in name and ("to_k" in name or "to_v" in name)


@torch.no_grad()
def calibrate(unet: nn.Module, batches, bos: bool = True) -> "OrderedDict[str, dict]":
    """Run `unet(**batch)` for every batch with hooks on each Linear/Conv2d input and return the
    kernel-format checkpoint.  `bos`: leave token 0 out of the cross-attention K/V statistics
    (it takes the FP16 carve-out, nn/Linear.py:178-194)."""
    from mixdq_amd.unet import quantizable_layers
    layers = quantizable_layers(unet)
    stats, hooks = {}, []

    def make_hook(name, mod):
        split = getattr(mod, "split", 0) if (
            "up_blocks" in name and "conv_shortcut" in name) else 0

        def hook(_m, args):
            x = args[0]
            if bos and _is_bos_layer(name) and x.dim() == 3 and x.shape[1] > 1:
                x = x[:, 1:, :]
            if split:
                stats.setdefault(name, ActRange()).update(x[:, :split])
                stats.setdefault(name + "#0", ActRange()).update(x[:, split:])
            else:
                stats.setdefault(name, ActRange()).update(x)
        return hook, split

    splits = {}
    for name, mod in layers.items():
        h, split = make_hook(name, mod)
        splits[name] = split
        hooks.append(mod.register_forward_pre_hook(h))
    try:
        for batch in batches:
            unet(**batch)
    finally:
        for h in hooks:
            h.remove()

    ckpt = OrderedDict()
    for name, mod in layers.items():
        split = splits[name]
        if split:
            ckpt[name + ".weight_quantizer"] = weight_quantizer_entry(mod.weight[:, :split])
            ckpt[name + ".weight_quantizer_0"] = weight_quantizer_entry(mod.weight[:, split:])
            ckpt[name + ".act_quantizer_0"] = stats[name + "#0"].entry()
        else:
            ckpt[name + ".weight_quantizer"] = weight_quantizer_entry(mod.weight)
        ckpt[name + ".act_quantizer"] = stats[name].entry()
    return ckpt


@torch.no_grad()
def precompute_bos(unet: nn.Module, encoder_hidden_states: torch.Tensor) -> dict:
    """bos_pre_computed.pt counterpart: the FP16 product of the first text token with each
    cross-attention to_k / to_v weight, shape [1, 1, C]."""
    from mixdq_amd.unet import quantizable_layers
    first = encoder_hidden_states[:1, :1, :]
    return {name: torch.nn.functional.linear(first, mod.weight)
            for name, mod in quantizable_layers(unet).items() if _is_bos_layer(name)}
