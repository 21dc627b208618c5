"""`quantize_unet(..., swap_glue=True)`: the stock glue modules between the quantized layers, swapped BY TYPE.

The reference's module swap replaces `nn.Linear` / `nn.Conv2d` only (quantize_sdxl.py:142-156, quantize.py:606-669);
everything between them stays stock PyTorch: `nn.GroupNorm` (+ `nn.SiLU`), `nn.LayerNorm`, the GEGLU activation and
`F.scaled_dot_product_attention` -- on MI355X 13.6 of the 27.3 ms of the swapped step (profiles/r05_dropin_glue.txt:
GroupNorm 46 x 69 us, SDPA 140 x 41 us, GELU + multiply 70 x 26 us).  This file extends the SAME monkey-patch surface
to those module types, each replaced by this repo's FP16-output kernel for that op -- unfused: one launch per module,
FP16 in, FP16 out, the quantize launch of the next layer untouched -- so the graph, the module names, the
state_dict keys and the call sites are those of the stock network:

    nn.GroupNorm                      -> HipGroupNorm    mixdq_groupnorm_silu_quantize3(out_f16)     csrc/fused_norm.hip
      (+ the nn.SiLU that follows it in a ResnetBlock2D / behind conv_norm_out: folded into that launch)
    nn.LayerNorm                      -> HipLayerNorm    mixdq_layernorm_quantize(out_f16)
    GEGLU (a module with `.proj`)     -> proj, then      mixdq_geglu_quantize(out_f16)
    the FP16 attention core           -> mixdq_attention_f16  (this repo's Attention.attend, or a diffusers
                                         attention processor: HipAttnProcessor)          csrc/attention.hip

The swap changes `module.__class__` to a subclass of the module's own class: parameters, buffers, hooks, names and
`isinstance` checks are untouched, `unswap_glue_modules` restores the stock classes.  A swapped module falls back to
its stock forward (PyTorch's op: still the GPU, never a CPU path) for inputs the kernel does not take (FP32, NCHW
memory, C % 16 != 0 ...).  Arithmetic: each kernel is within one FP16 ulp per rounding point of PyTorch's
FP32-reference op and bit-equal to the fused graph's `out_h` of the same kernel (tests/test_glue_gpu.py).
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

_ACT_TAG = "_mixdq_silu_applied"


def _f16_cuda(x) -> bool:
    return torch.is_tensor(x) and x.is_cuda and x.dtype == torch.float16


class HipGroupNorm(nn.GroupNorm):
    """nn.GroupNorm on the HIP kernel (statistics -> finalize -> apply, fixed reduction order).  `fuse_silu`: the
    parent applies an nn.SiLU to this module's output and to nothing else of it; the activation then runs in the
    apply pass and the output is tagged so that the (swapped) nn.SiLU module passes it through."""
    fuse_silu = False

    def forward(self, x):
        from mixdq_amd import _C
        if (_f16_cuda(x) and x.dim() == 4 and self.affine and self.weight.dtype == torch.float16
                and x.is_contiguous(memory_format=torch.channels_last)
                and _C.groupnorm_supported(x.shape[0], x.shape[2] * x.shape[3], x.shape[1], self.num_groups)):
            y = _C.groupnorm_silu_quantize(x, self.num_groups, self.weight, self.bias, self.eps,
                                           silu=self.fuse_silu, want_f16=True)[1]
            if self.fuse_silu:
                setattr(y, _ACT_TAG, True)
            return y
        return super().forward(x)


class HipSiLU(nn.SiLU):
    """nn.SiLU that passes through a tensor whose activation already ran in HipGroupNorm's launch."""

    def forward(self, x):
        if getattr(x, _ACT_TAG, False):
            return x
        return super().forward(x)


class HipLayerNorm(nn.LayerNorm):
    def forward(self, x):
        from mixdq_amd import _C
        C = x.shape[-1] if torch.is_tensor(x) and x.dim() else 0
        if (_f16_cuda(x) and x.is_contiguous() and len(self.normalized_shape) == 1 and self.elementwise_affine
                and self.bias is not None and self.weight.dtype == torch.float16
                and C == self.normalized_shape[0] and C % 16 == 0 and 0 < C <= 2048):
            return _C.layernorm_quantize(x, self.weight, self.bias, self.eps, [], want_f16=True)[1]
        return super().forward(x)


class _HipGEGLU:
    """Mix-in for a GEGLU module (`self.proj`: Linear(dim, 2 * inner); forward = value * gelu(gate) of its two
    halves -- mixdq_amd.unet.GEGLU, diffusers.models.activations.GEGLU): the chunk, the GELU and the multiply
    (three PyTorch kernels and two strided reads of the projection) are one launch."""

    def forward(self, x, *args, **kwargs):
        from mixdq_amd import _C
        h = self.proj(x)
        if _f16_cuda(h) and h.is_contiguous() and h.shape[-1] % 16 == 0:
            return _C.geglu_quantize(h, want_f16=True)[1]
        v, g = h.chunk(2, dim=-1)
        return v * F.gelu(g)


def _attention_core(q, k, v, heads):
    """FP16 attention on [B, T, C] tensors (heads of 64 columns), or None where the kernel does not take them."""
    from mixdq_amd import _C
    C = q.shape[-1]
    if C != heads * 64:
        return None
    for t in (q, k, v):
        if not (_f16_cuda(t) and t.dim() == 3 and t.stride(-1) == 1 and t.stride(0) % 8 == 0
                and t.stride(1) % 8 == 0 and t.data_ptr() % 16 == 0):
            return None
    return _C.attention_f16(q, k, v, heads)


class _HipAttend:
    """Mix-in for mixdq_amd.unet.Attention: `attend` (the FP16 core between the projections) on the HIP kernel."""

    def attend(self, q, k, v):
        o = _attention_core(q, k, v, self.heads)
        return o if o is not None else super().attend(q, k, v)


class HipAttnProcessor:
    """A diffusers attention processor (the `__call__(attn, hidden_states, encoder_hidden_states, ...)` interface of
    AttnProcessor2_0) whose softmax(q k^T) v runs on mixdq_attention_f16; the projections stay the (quantized)
    modules of `attn`.  Installed by `attn.set_processor(HipAttnProcessor())` for every module that has one.
    Covers what the SDXL UNet uses: no attention mask, no group / spatial norm inside the attention, no added
    key / value projections, residual_connection False -- anything else goes to the processor it replaced."""

    def __init__(self, fallback=None):
        self.fallback = fallback

    def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None, temb=None,
                 *args, **kwargs):
        plain = (attention_mask is None and hidden_states.dim() == 3
                 and getattr(attn, "group_norm", None) is None and getattr(attn, "spatial_norm", None) is None
                 and getattr(attn, "norm_q", None) is None and getattr(attn, "norm_k", None) is None
                 and not getattr(attn, "residual_connection", False)
                 and getattr(attn, "rescale_output_factor", 1.0) == 1.0)
        if plain:
            ctx = hidden_states if encoder_hidden_states is None else encoder_hidden_states
            if encoder_hidden_states is not None and getattr(attn, "norm_cross", False):
                ctx = attn.norm_encoder_hidden_states(ctx)
            q, k, v = attn.to_q(hidden_states), attn.to_k(ctx), attn.to_v(ctx)
            o = _attention_core(q, k, v, attn.heads)
            if o is not None:
                o = attn.to_out[0](o)
                return attn.to_out[1](o) if len(attn.to_out) > 1 else o
        if self.fallback is None:
            raise RuntimeError("HipAttnProcessor: unsupported attention call and no fallback processor")
        return self.fallback(attn, hidden_states, encoder_hidden_states=encoder_hidden_states,
                             attention_mask=attention_mask, temb=temb, *args, **kwargs)


def _subclass(mixin, base):
    """A class `Hip<base>` = (mixin, base), cached per base class."""
    cache = _subclass.__dict__.setdefault("cache", {})
    key = (mixin, base)
    if key not in cache:
        cache[key] = type("Hip" + base.__name__, (mixin, base), {"_mixdq_stock_class": base})
    return cache[key]


def _is_geglu(mod) -> bool:
    return type(mod).__name__ == "GEGLU" and isinstance(getattr(mod, "proj", None), nn.Module)


# Parent classes (by name) whose forward is KNOWN to apply the nn.SiLU module `act` to the output of each GroupNorm in
# `norms`, and to nothing else of it: diffusers' and this repo's ResnetBlock2D (norm1 / norm2 -> nonlinearity) and the
# UNet itself (conv_norm_out -> conv_act).  Only there is the SiLU folded into the GroupNorm's launch: a parent that
# applied F.silu itself, or used the GroupNorm's output twice, would get the activation twice or in the wrong place.
SILU_PAIRS = {"ResnetBlock2D": (("norm1", "norm2"), "nonlinearity"),
              "UNet2DConditionModel": (("conv_norm_out",), "conv_act"),
              "SDXLUNet": (("conv_norm_out",), "conv_act")}


def swap_glue_modules(unet: nn.Module, attention: bool = True, silu_pairs=None) -> dict:
    """Swap the stock glue modules of `unet` in place (see the top of this file); returns how many of each kind
    were swapped.  Idempotent.  `attention=False` leaves the attention core to PyTorch's SDPA.  `silu_pairs`:
    {parent class name: ((GroupNorm attribute names), SiLU attribute name)} in place of SILU_PAIRS."""
    n = dict(groupnorm=0, silu_folded=0, layernorm=0, geglu=0, attention=0)
    pairs = SILU_PAIRS if silu_pairs is None else silu_pairs
    for parent in unet.modules():
        kids = parent._modules
        spec = pairs.get(type(parent).__name__)
        for norms, act in ((spec,) if spec else ()):
            a = kids.get(act)
            if a is None or type(a) not in (nn.SiLU, HipSiLU):
                continue
            for nm in norms:
                g = kids.get(nm)
                if g is not None and type(g) in (nn.GroupNorm, HipGroupNorm) and not g.__dict__.get("fuse_silu"):
                    g.__dict__["fuse_silu"] = True
                    n["silu_folded"] += 1
            if type(a) is nn.SiLU:
                a.__class__ = HipSiLU
    for mod in unet.modules():
        if type(mod) is nn.GroupNorm:
            mod.__class__ = HipGroupNorm
            n["groupnorm"] += 1
        elif type(mod) is nn.LayerNorm:
            mod.__class__ = HipLayerNorm
            n["layernorm"] += 1
        elif _is_geglu(mod) and not isinstance(mod, _HipGEGLU):
            mod.__class__ = _subclass(_HipGEGLU, type(mod))
            n["geglu"] += 1
        elif attention and not isinstance(mod, _HipAttend) and hasattr(type(mod), "attend") \
                and hasattr(mod, "heads") and hasattr(mod, "to_q"):
            mod.__class__ = _subclass(_HipAttend, type(mod))
            n["attention"] += 1
        elif attention and hasattr(mod, "set_processor") and hasattr(mod, "to_q") \
                and not isinstance(getattr(mod, "processor", None), HipAttnProcessor):
            mod.set_processor(HipAttnProcessor(fallback=getattr(mod, "processor", None)))
            n["attention"] += 1
    return n


def unswap_glue_modules(unet: nn.Module) -> None:
    """Restore the stock classes (and attention processors) swap_glue_modules replaced."""
    for mod in unet.modules():
        if type(mod) is HipGroupNorm:
            mod.__dict__.pop("fuse_silu", None)
            mod.__class__ = nn.GroupNorm
        elif type(mod) is HipSiLU:
            mod.__class__ = nn.SiLU
        elif type(mod) is HipLayerNorm:
            mod.__class__ = nn.LayerNorm
        elif hasattr(type(mod), "_mixdq_stock_class"):
            mod.__class__ = type(mod)._mixdq_stock_class
        elif isinstance(getattr(mod, "processor", None), HipAttnProcessor) and mod.processor.fallback is not None:
            mod.set_processor(mod.processor.fallback)
