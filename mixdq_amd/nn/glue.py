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

Operand hand-off (`operands=True`, the default): a swapped producer whose parent class is known to pass its output
straight to quantized layers (OPERAND_PAIRS: LayerNorm -> to_q / to_k / to_v / the GEGLU projection, GroupNorm + SiLU ->
the ResNet conv, GEGLU -> ff.net.2; the attention core -> to_out.0 directly) also writes, in the SAME launch, the INT8
operand those layers' own quantize launch would compute from its FP16 output -- quantize() of the FP16 value, the same
bits -- and attaches it to the FP16 tensor it returns.  QuantizedLinear / QuantizedConv2d look for an attachment made
with THEIR quantizer tensors on THEIR input tensor object (`tagged_operand`: identity of the tensor, of the two
quantizer buffers, and the tensor's in-place version) and skip the quantize launch; anything else -- another tensor, a
modified one, a re-quantized layer -- finds nothing and quantizes as before.  The FP16 tensor is always written too.
Inside a swapped attention module, where nothing escapes between its input and to_out.0's output
(`_attention_hand_off`): to_q / to_k / to_v of a self-attention are ONE GEMM against their row-concatenated weights,
the attention launch writes to_out.0's INT8 operand, a cross-attention's k / v go into kept BOS buffers from one shared
quantize launch, and to_q + attention + to_out.0's quantizer are one launch -- the launch forms of the fused graph,
each bit-identical to the chain it replaces (tests/test_fused_gpu.py).

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
_OPS_TAG = "_mixdq_operands"        # on an FP16 tensor: (its _version, [(scale_inv, zero_point, int8 tensor), ...])
_CONSUMERS = "_mixdq_consumers"     # in a swapped producer's __dict__: the layers its parent hands its output to


def _f16_cuda(x) -> bool:
    return torch.is_tensor(x) and x.is_cuda and x.dtype == torch.float16


def _takes_operand(m) -> bool:
    """A W8A8 / W4A8 layer that quantizes its whole input with ONE per-tensor quantizer (no BOS carve-out, no
    split input)."""
    return bool(getattr(m, "valid_for_acceleration", False) and not getattr(m, "bos", False)
                and not getattr(m, "split", 0) and hasattr(m, "act_scales_inv"))


def _consumers(mod, width, device):
    """The linked layers that would quantize a [.., width] tensor on `device` with a quantizer that lives there."""
    return [c for c in mod.__dict__.get(_CONSUMERS, ())
            if _takes_operand(c) and getattr(c, "in_features", getattr(c, "in_channels", None)) == width
            and c.act_scales_inv.device == device and c.act_zero_points.device == device]


def _attach(y, consumers, ints):
    """`ints[i]`: quantize(y) with consumers[i]'s quantizer (or None)."""
    ops = [(c.act_scales_inv, c.act_zero_points, q) for c, q in zip(consumers, ints) if q is not None]
    if ops:
        setattr(y, _OPS_TAG, (y._version, ops))
    return y


def tagged_operand(x, layer):
    """The INT8 operand a swapped producer attached to `x` for `layer`'s activation quantizer, or None."""
    tag = getattr(x, _OPS_TAG, None)
    if tag is None or tag[0] != x._version:
        return None
    s, z = layer.act_scales_inv, layer.act_zero_points
    for ts, tz, q in tag[1]:
        if ts is s and tz is z and q.shape == x.shape and q.device == x.device:
            return q
    return None


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
            # (the consumer reads silu(norm(x)): its operand can ride along only where the SiLU does)
            cons = _consumers(self, x.shape[1], x.device)[:1] if self.fuse_silu else []
            qp = (cons[0].act_scales_inv, cons[0].act_zero_points) if cons else (None, None)
            q, y = _C.groupnorm_silu_quantize(x, self.num_groups, self.weight, self.bias, self.eps, *qp,
                                              silu=self.fuse_silu, want_f16=True)[:2]
            if self.fuse_silu:
                setattr(y, _ACT_TAG, True)
            return _attach(y, cons, [q])
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
            cons = _consumers(self, C, x.device)
            if not cons:
                return _C.layernorm_quantize(x, self.weight, self.bias, self.eps, [], want_f16=True)[1]
            # one INT8 tensor per DISTINCT quantizer among the consumers (to_q / to_k / to_v are calibrated on the
            # same tensor: usually one); at most three ride in the launch
            from mixdq_amd.unet import _quantizer_groups
            ids = _quantizer_groups(self.__dict__.setdefault("_mixdq_memo", {}), "glue", cons)
            first = {}
            for c, i in zip(cons, ids):
                first.setdefault(i, c)
            order = sorted(first)[:3]
            outs, y = _C.layernorm_quantize(x, self.weight, self.bias, self.eps,
                                            [(first[i].act_scales_inv, first[i].act_zero_points) for i in order],
                                            want_f16=True)
            return _attach(y, cons, [outs[order.index(i)] if i in order else None for i in ids])
        return super().forward(x)


class _HipGEGLU:
    """Mix-in for a GEGLU module (`self.proj`: Linear(dim, 2 * inner); forward = value * gelu(gate) of its two
    halves -- mixdq_amd.unet.GEGLU, diffusers.models.activations.GEGLU): the chunk, the GELU and the multiply
    (three PyTorch kernels and two strided reads of the projection) are one launch."""

    def forward(self, x, *args, **kwargs):
        from mixdq_amd import _C
        h = self.proj(x)
        if _f16_cuda(h) and h.is_contiguous() and h.shape[-1] % 16 == 0:
            cons = _consumers(self, h.shape[-1] // 2, h.device)[:1]
            qp = (cons[0].act_scales_inv, cons[0].act_zero_points) if cons else (None, None)
            q, y = _C.geglu_quantize(h, *qp, want_f16=True)
            return _attach(y, cons, [q])
        v, g = h.chunk(2, dim=-1)
        return v * F.gelu(g)


_BOS_BUFS = "_mixdq_bos_out"          # in a BOS layer's __dict__: {(B, T, device): [buffer, bos row tensor, its version]}
# A kept buffer is never freed while its layer lives (a captured graph may hold its address), so their NUMBER is
# bounded instead: a layer that has handed out this many (batch, tokens, device) shapes serves further shapes the
# reference's way (allocate + copy row 0 per call).  197 KB per image at 77 x 1280: 16 shapes of <= 64 images stay
# under 0.2 GB per layer pair even in the worst case; a service with a few fixed batch sizes never gets near it.
BOS_BUFFERS_MAX = int(__import__("os").environ.get("MIXDQ_BOS_BUFFERS_MAX", "16"))


def _project_context(layer, ctx):
    """`layer(ctx)` for the key / value projection of a swapped attention.  A BOS layer (token 0 of the text context
    is a precomputed FP16 row, nn/Linear.py) writes into a buffer the layer keeps per (batch, tokens, device), whose
    row 0 was filled once -- the reference's `out[:, :1] = bos_pre_computed` copy per call (140 launches per step)
    disappears.  Safe HERE because the swapped attention consumes k / v before it returns; every buffer ever handed
    out stays alive with the layer (a captured hipGraph holds its address), and row 0 is re-filled in place when
    the BOS row changes (load_state_dict: the buffer's in-place version)."""
    if not _is_bos_layer(layer, ctx):
        return layer(ctx)
    return layer(ctx, _bos_out=_bos_buffer(layer, ctx))        # (None past BOS_BUFFERS_MAX shapes: the plain call)


def _bos_buffer(layer, ctx):
    """The kept [B, T, N] output buffer of a BOS layer for this (batch, tokens, device), row 0 filled; None when the
    layer already keeps BOS_BUFFERS_MAX of them.  One forward at a time per module, as with a captured graph."""
    bufs = layer.__dict__.setdefault(_BOS_BUFS, {})
    key = (ctx.shape[0], ctx.shape[1], ctx.device)
    row = layer.bos_pre_computed
    e = bufs.get(key)
    if e is None:
        if len(bufs) >= BOS_BUFFERS_MAX:
            return None
        e = bufs[key] = [torch.empty((ctx.shape[0], ctx.shape[1], layer.out_features), dtype=torch.float16,
                                     device=ctx.device), None, -1]
    if e[1] is not row or e[2] != row._version:
        with torch.no_grad():
            e[0][:, :1, :] = row
        e[1], e[2] = row, row._version
    return e[0]


def _is_bos_layer(layer, ctx) -> bool:
    return bool(getattr(layer, "valid_for_acceleration", False) and getattr(layer, "bos", False)
                and _f16_cuda(ctx) and ctx.dim() == 3 and ctx.shape[1] > 1
                and torch.is_tensor(getattr(layer, "bos_pre_computed", None))
                and layer.bos_pre_computed.device == ctx.device and layer.act_scales_inv.device == ctx.device)


def _project_kv(to_k, to_v, ctx):
    """(to_k(ctx), to_v(ctx)) for a swapped attention.  Two BOS layers with EQUAL activation quantizers (they are
    calibrated on the same tensor) share one quantize launch of the context's tokens 1.. -- the reference runs it
    once per layer -- unless the measured table prefers the quantizing GEMM for the shape."""
    if _is_bos_layer(to_k, ctx) and _is_bos_layer(to_v, ctx) and to_k.in_features == to_v.in_features:
        from mixdq_amd import _C
        from mixdq_amd.unet import _quantizer_groups
        ids = _quantizer_groups(to_k.__dict__.setdefault("_mixdq_memo", {}), "glue_kv", [to_k, to_v])
        if ids[0] == ids[1] and not any(
                _C.qlinear_f16in_wanted(ctx, m.out_features, m.in_features, w4=m.w_packed4, bos=True)
                for m in (to_k, to_v)):
            B, T = ctx.shape[0], ctx.shape[1]
            from mixdq_amd.op.quant import quantize_per_tensor_vectorized as quant_op
            x_int = quant_op(ctx[:, 1:, :], to_k.act_scales_inv, to_k.act_zero_points)
            # (out=None: forward_bos_quantized allocates and fills row 0 itself, as the reference does)
            return (to_k.forward_bos_quantized(x_int, B, T, out=_bos_buffer(to_k, ctx)),
                    to_v.forward_bos_quantized(x_int, B, T, out=_bos_buffer(to_v, ctx)))
    return _project_context(to_k, ctx), _project_context(to_v, ctx)


def _attention_core(q, k, v, heads, out_layer=None):
    """FP16 attention on [B, T, C] tensors (heads of 64 columns), or None where the kernel does not take them.
    `out_layer` (to_out.0): where it is a quantized layer that takes an operand, the launch writes that layer's
    INT8 operand instead of the FP16 tensor and the return value is the layer's OUTPUT."""
    from mixdq_amd import _C
    C = q.shape[-1]
    if C != heads * 64:
        return None
    for t in (q, k, v):
        if not (_f16_cuda(t) and t.dim() == 3 and t.stride(-1) == 1 and t.stride(0) % 8 == 0
                and t.stride(1) % 8 == 0 and t.data_ptr() % 16 == 0):
            return None
    if out_layer is not None:
        if (_takes_operand(out_layer) and getattr(out_layer, "in_features", None) == C
                and out_layer.act_scales_inv.device == q.device):
            return out_layer._gemm(_C.attention_f16(q, k, v, heads, out_layer.act_scales_inv,
                                                    out_layer.act_zero_points))
        return out_layer(_C.attention_f16(q, k, v, heads))
    return _C.attention_f16(q, k, v, heads)


def _operand(x, layer):
    """quantize(x) with `layer`'s quantizer: the one its producer attached, else the layer's own quantize launch."""
    q = tagged_operand(x, layer)
    if q is None:
        from mixdq_amd.op.quant import quantize_per_tensor_vectorized as quant_op
        q = quant_op(x, layer.act_scales_inv, layer.act_zero_points)
    return q


def _self_qkv(attn, x):
    """(q, k, v) of a self-attention from ONE GEMM against the row-concatenated weights of to_q / to_k / to_v --
    three W8A8 (or three packed 4-bit) layers without bias whose activation quantizers are equal (they read the same
    tensor): per-channel scale / bias0 concatenate, so every output element is computed exactly as by the three
    launches (mixdq_amd.unet._pack_rows: the pack is the storage, the layers' buffers become views of it -- names,
    shapes and values of the state_dict are unchanged).  None where the layers do not qualify."""
    layers = [attn.to_q, attn.to_k, attn.to_v]
    C = x.shape[-1]
    if not (x.is_contiguous() and all(_takes_operand(m) and m.bias is None and m.in_features == C
                                      and m.act_scales_inv.device == x.device for m in layers)):
        return None
    from mixdq_amd.op.qlinear import qlinear
    from mixdq_amd.unet import _pack_rows, _pack_valid, _quantizer_groups, _uniform_storage
    if not _uniform_storage(layers) or len(set(_quantizer_groups(
            attn.__dict__.setdefault("_mixdq_memo", {}), "glue_qkv", layers))) != 1:
        return None
    pack = attn.__dict__.get("_qkv")
    if not _pack_valid(pack, layers):
        pack = attn.__dict__["_qkv"] = _pack_rows(layers)
    q0 = layers[0]
    x_int = next((t for t in (tagged_operand(x, m) for m in layers) if t is not None), None)
    if x_int is None:
        x_int = _operand(x, q0)
    qkv = qlinear(x_int, pack["w"], pack["wscale"], q0.act_scales, q0.act_zero_points, pack["wsum"], pack["scale"],
                  pack["bias0"], None, _w4=pack["w4"])
    n = pack["C"]
    return qkv[..., :n], qkv[..., n:2 * n], qkv[..., 2 * n:]


def _cross_one_launch(attn, x, k, v):
    """to_out.0(attention(to_q(x), k, v)) of a cross-attention with to_q's GEMM, the attention core and to_out.0's
    quantizer in ONE launch (mixdq_qlinear_w8a8_attn: the 77 keys / values fit a workgroup's LDS; bit-identical to
    the three steps, tests/test_fused_gpu.py).  None where the launch does not take the problem."""
    from mixdq_amd import _C
    from mixdq_amd.unet import CROSS_FUSE_MAX_ROWS
    q, out = attn.to_q, attn.to_out[0]
    if not (_takes_operand(q) and q.bias is None and _takes_operand(out) and x.is_contiguous()
            and q.in_features == x.shape[-1] and attn.heads * 64 == q.out_features == out.in_features
            and q.act_scales_inv.device == x.device and out.act_scales_inv.device == x.device
            and all(_f16_cuda(z) and z.dim() == 3 and z.stride(-1) == 1 and z.stride(0) % 8 == 0
                    and z.stride(1) % 8 == 0 and z.data_ptr() % 16 == 0 for z in (k, v))
            and k.shape == v.shape and k.shape[0] == x.shape[0]
            and x.shape[0] * x.shape[1] <= CROSS_FUSE_MAX_ROWS
            and _C.qlinear_attention_supported(x.shape, q.out_features, q.in_features, k)):
        return None
    o_int = _C.qlinear_attention(_operand(x, q), q.weight_int4 if q.w_packed4 else q.weight_int, q.scale, q.bias0,
                                 k, v, out.act_scales_inv, out.act_zero_points, _w4=q.w_packed4)
    return out._gemm(o_int)


def _attention_hand_off(attn, x, context):
    """to_out.0(attention(...)) of a swapped attention module with everything between its input and to_out.0's
    output inside this function (nothing escapes: kept buffers and INT8 intermediates are safe): self-attention =
    one q | k | v GEMM + the attention launch writing to_out.0's operand; cross-attention = k / v into kept BOS
    buffers from one shared quantize launch + (to_q, attention, to_out.0's quantizer) in one launch.  Each step
    falls back to the module-by-module form where its conditions do not hold; None: not an input of the kernels."""
    if not (_f16_cuda(x) and x.dim() == 3 and getattr(attn.to_q, "out_features", None) == attn.heads * 64):
        return None                                     # (decided before anything is launched)
    if context is None:
        qkv = _self_qkv(attn, x)
        q, k, v = qkv if qkv is not None else (attn.to_q(x), attn.to_k(x), attn.to_v(x))
    else:
        k, v = _project_kv(attn.to_k, attn.to_v, context)
        y = _cross_one_launch(attn, x, k, v)
        if y is not None:
            return y
        q = attn.to_q(x)
    y = _attention_core(q, k, v, attn.heads, out_layer=attn.to_out[0])
    if y is None:         # (operands the kernel does not take after all: PyTorch's core on the projections made)
        B, T, C = q.shape
        h = attn.heads
        o = F.scaled_dot_product_attention(*(t.unflatten(-1, (h, C // h)).transpose(1, 2) for t in (q, k, v)))
        y = attn.to_out[0](o.transpose(1, 2).reshape(B, T, C))
    return y


class _HipAttend:
    """Mix-in for mixdq_amd.unet.Attention: `attend` (the FP16 core between the projections) on the HIP kernel."""
    hand_off = False            # (swap_glue_modules(operands=True): see _attention_hand_off)

    def attend(self, q, k, v):
        o = _attention_core(q, k, v, self.heads)
        return o if o is not None else super().attend(q, k, v)

    def forward(self, x, context=None):
        y = _attention_hand_off(self, x, context) if self.hand_off else None
        return y if y is not None else super().forward(x, context)


class HipAttnProcessor:
    """A diffusers attention processor (the `__call__(attn, hidden_states, encoder_hidden_states, ...)` interface of
    AttnProcessor2_0) whose softmax(q k^T) v runs on mixdq_attention_f16; the projections stay the (quantized)
    modules of `attn`.  Installed by `attn.set_processor(HipAttnProcessor())` for every module that has one.
    Covers what the SDXL UNet uses: no attention mask, no group / spatial norm inside the attention, no added
    key / value projections, residual_connection False, softmax scale 1 / 8 -- anything else goes to the processor
    it replaced."""

    def __init__(self, fallback=None, hand_off=False):
        self.fallback = fallback
        self.hand_off = hand_off    # to_out.0's INT8 operand written by the attention launch

    def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None, temb=None,
                 *args, **kwargs):
        plain = (attention_mask is None and hidden_states.dim() == 3
                 and getattr(attn, "group_norm", None) is None and getattr(attn, "spatial_norm", None) is None
                 and getattr(attn, "norm_q", None) is None and getattr(attn, "norm_k", None) is None
                 and not getattr(attn, "residual_connection", False)
                 and getattr(attn, "rescale_output_factor", 1.0) == 1.0
                 and getattr(attn, "scale", 0.125) == 0.125)       # (the kernel's softmax scale: 64 ** -0.5)
        if plain:
            ctx = hidden_states if encoder_hidden_states is None else encoder_hidden_states
            if encoder_hidden_states is not None and getattr(attn, "norm_cross", False):
                ctx = attn.norm_encoder_hidden_states(ctx)
            o = None
            if self.hand_off:
                o = _attention_hand_off(attn, hidden_states, None if encoder_hidden_states is None else ctx)
            if o is None:
                o = _attention_core(attn.to_q(hidden_states), attn.to_k(ctx), attn.to_v(ctx), attn.heads)
                if o is not None:
                    o = attn.to_out[0](o)
            if o is not None:
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


# Parent classes (by name) whose forward is KNOWN to hand the output of the producer at the first path to the layers
# at the other paths, as the same tensor object (through at most nn.SiLU / nn.Dropout / nn.Identity modules, which
# pass an eval-mode tensor through): diffusers' and this repo's BasicTransformerBlock, ResnetBlock2D and FeedForward.
# A wrong entry costs time, never bits: a consumer only ever uses an operand attached to ITS input tensor for ITS
# quantizer (`tagged_operand`).
OPERAND_PAIRS = {
    "BasicTransformerBlock": (("norm1", ("attn1.to_q", "attn1.to_k", "attn1.to_v")),
                              ("norm2", ("attn2.to_q",)),
                              ("norm3", ("ff.net.0.proj",))),
    "ResnetBlock2D": (("norm1", ("conv1",)), ("norm2", ("conv2",))),
    "FeedForward": (("net.0", ("net.2",)),),
}


def _submodule(root, path):
    for part in path.split("."):
        root = getattr(root, "_modules", {}).get(part)
        if root is None:
            return None
    return root


def swap_glue_modules(unet: nn.Module, attention: bool = True, silu_pairs=None, operands: bool = True,
                      operand_pairs=None) -> dict:
    """Swap the stock glue modules of `unet` in place (see the top of this file); returns how many of each kind
    were swapped.  Idempotent.  `attention=False` leaves the attention core to PyTorch's SDPA.  `silu_pairs`:
    {parent class name: ((GroupNorm attribute names), SiLU attribute name)} in place of SILU_PAIRS.
    `operands=False`: no INT8 operand hand-off between swapped producers and the quantized layers behind them
    (every layer runs its own quantize launch, as in the reference); `operand_pairs` in place of OPERAND_PAIRS."""
    n = dict(groupnorm=0, silu_folded=0, layernorm=0, geglu=0, attention=0, operand_links=0, attention_handoff=0)
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
    if not operands:
        return n
    opairs = OPERAND_PAIRS if operand_pairs is None else operand_pairs
    for parent in unet.modules():
        for prod_path, cons_paths in opairs.get(type(parent).__name__, ()):
            prod = _submodule(parent, prod_path)
            if not isinstance(prod, (HipGroupNorm, HipLayerNorm, _HipGEGLU)) or _CONSUMERS in prod.__dict__:
                continue
            cons = tuple(c for c in (_submodule(parent, p) for p in cons_paths) if c is not None)
            if cons:
                prod.__dict__[_CONSUMERS] = cons       # (not registered as sub-modules: plain references)
                n["operand_links"] += len(cons)
    for mod in unet.modules():
        if isinstance(mod, _HipAttend) and not mod.__dict__.get("hand_off"):
            mod.__dict__["hand_off"] = True
            n["attention_handoff"] += 1
        elif isinstance(getattr(mod, "processor", None), HipAttnProcessor) and not mod.processor.hand_off:
            mod.processor.hand_off = True
            n["attention_handoff"] += 1
    return n


def unswap_glue_modules(unet: nn.Module) -> None:
    """Restore the stock classes (and attention processors) swap_glue_modules replaced."""
    for mod in unet.modules():
        mod.__dict__.pop(_CONSUMERS, None)
        mod.__dict__.pop("hand_off", None)
        mod.__dict__.pop(_BOS_BUFS, None)
        if isinstance(mod, _HipAttend) or isinstance(getattr(mod, "processor", None), HipAttnProcessor):
            mod.__dict__.pop("_qkv", None)      # (_self_qkv's pack record; the layers' buffers stay views of the pack)
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
