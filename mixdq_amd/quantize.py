"""Module swap: counterpart of kernels/quantize.py:527-669 (convert / _convert / swap_module).

The reference forks torch.ao.quantization.convert to thread a `ckpt` and a `split` through to
`from_float`.  Here the walk is written directly: every sub-module that carries a `.qconfig` and
whose type is in `mapping` is replaced by `mapping[type].from_float(mod, split=..., ckpt=ckpt)`;
forward hooks and device placement are preserved as in swap_module (quantize.py:651-668).

Split widths of the nine up-block conv_shortcuts: the reference keeps them in a module-global
list consumed by a global counter (quantize.py:61-64,631-643), so a second convert() in one
process raises IndexError.  Here the sequence is local to one convert() call, and a module may
state its own split (`mod.split`, set by mixdq_amd.unet's up-block resnets).
"""
from __future__ import annotations

import copy

import torch.nn as nn

# hidden-state channel count of the up-block resnets, in named_modules() order (quantize.py:61)
SDXL_UP_SHORTCUT_SPLITS = (1280, 1280, 1280, 1280, 640, 640, 640, 320, 320)


class _SplitSequence:
    def __init__(self, splits):
        self._splits = tuple(splits)
        self._next = 0

    def take(self, mod):
        own = getattr(mod, "split", None)
        if isinstance(own, int) and own > 0:
            self._next += 1
            return own
        if self._next >= len(self._splits):
            raise IndexError(
                f"more up-block conv_shortcut layers than split widths ({len(self._splits)})")
        s = self._splits[self._next]
        self._next += 1
        return s


def _is_split_shortcut(name: str) -> bool:
    return "up_blocks" in name and "conv_shortcut" in name


def swap_module(mod, mapping, ckpt=None, splits: _SplitSequence | None = None):
    """Return the quantized counterpart of `mod` if it has a qconfig and a mapped type."""
    if getattr(mod, "qconfig", None) is None or type(mod) not in mapping:
        return mod
    split = 0
    if _is_split_shortcut(getattr(mod, "module_name", "")):
        split = (splits or _SplitSequence(SDXL_UP_SHORTCUT_SPLITS)).take(mod)
    new_mod = mapping[type(mod)].from_float(mod, split=split, ckpt=ckpt)
    for hook in mod._forward_pre_hooks.values():
        new_mod.register_forward_pre_hook(hook)
    for hook in mod._forward_hooks.values():
        new_mod.register_forward_hook(hook)
    devices = {p.device for p in mod.parameters()} | {b.device for b in mod.buffers()}
    assert len(devices) <= 1, f"swap_module needs a single-device module, got {devices}"
    if devices:
        new_mod.to(next(iter(devices)))
    return new_mod


def _convert(module, mapping, ckpt, splits):
    for name, child in list(module.named_children()):
        _convert(child, mapping, ckpt, splits)
        module._modules[name] = swap_module(child, mapping, ckpt=ckpt, splits=splits)
    return module


def _remove_qconfig(module):
    for m in module.modules():
        if hasattr(m, "qconfig"):
            del m.qconfig


def convert(module, mapping=None, inplace=False, remove_qconfig=True, ckpt=None,
            splits=SDXL_UP_SHORTCUT_SPLITS):
    """Swap every qconfig-carrying nn.Linear / nn.Conv2d for its quantized counterpart."""
    if mapping is None:
        from mixdq_amd.nn import QuantizedConv2d, QuantizedLinear
        mapping = {nn.Linear: QuantizedLinear, nn.Conv2d: QuantizedConv2d}
    if not inplace:
        module = copy.deepcopy(module)
    _convert(module, mapping, ckpt, _SplitSequence(splits))
    if remove_qconfig:
        _remove_qconfig(module)
    return module
