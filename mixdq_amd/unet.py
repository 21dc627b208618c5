"""SDXL(-Turbo) UNet graph with diffusers-compatible module names.

diffusers is not installed anywhere this repo runs, and the reference's harness
(kernels/quantize_sdxl.py:331-484) drives `pipeline.unet`.  This module is a structural clone of
`UNet2DConditionModel` for the SDXL config (block channels 320/640/1280, transformer depth
0/2/10 + mid 10, cross-attention dim 2048, text_time addition embedding 2816 -> 1280) whose
`named_modules()` reproduce exactly the 794 quantizable layer names of the reference's bit-width
yamls (743 Linear + 51 Conv2d), so `quantize_unet` drops in unchanged.
Every Linear/Conv2d is a plain nn.Linear/nn.Conv2d until `mixdq_amd.quantize.convert` swaps it.

The glue between quantized layers (GroupNorm, SiLU, LayerNorm, GEGLU, scaled-dot-product
attention, nearest upsample) stays FP16 PyTorch, as in the reference (SURVEY.md section 0:
the reference fuses none of it and runs attention in FP16).  Activations are kept channels-last
so the INT8 NHWC conv never converts layouts (qconv2d.cc:91-95 would copy).

Weights are synthetic (no network, no checkpoints): randn * 0.02, seeded per layer.
"""
from __future__ import annotations

import math
import weakref
from collections import OrderedDict

import torch
import torch.nn as nn
import torch.nn.functional as F

# ---------------------------------------------------------------------------------------------
# Fused forward.  With `unet.fused = True` (set_fused) the glue that
# feeds a W8A8 layer is replaced by the producer-fusion kernels of mixdq_amd/csrc/fused_norm.hip
# and the residual adds move into the GEMM / conv epilogues.  Rounding points are those of the
# unfused graph; the only numerical difference is this repo's GroupNorm / LayerNorm / SiLU / GELU
# arithmetic (include/mixdq_math.h, fixed reduction order) instead of PyTorch's, i.e. at most
# one FP16 ulp before quantization.  Off by default: the unfused graph is the drop-in path.
# ---------------------------------------------------------------------------------------------
# De-fused reference of the fused graph (tests/test_unet_full_gpu.py): with DEFUSE on, the fused
# forward keeps its structure but every fused launch is replaced by the chain of THIS repo's
# kernels it stands for -- GroupNorm / LayerNorm / GEGLU / attention with FP16 output followed by
# the layer's own quantize launch, separate to_q / to_k / to_v and K / V GEMMs, torch half adds in
# place of the residual epilogues, torch.cat in place of the two-source reads.  Same arithmetic at
# every rounding point, so the two graphs agree BIT FOR BIT and any difference is a wiring mistake.
DEFUSE = False


class defused:
    """`with defused(): unet(...)` runs the de-fused reference of the fused graph."""

    def __enter__(self):
        global DEFUSE
        self.saved, DEFUSE = DEFUSE, True

    def __exit__(self, *exc):
        global DEFUSE
        DEFUSE = self.saved


def _accel(m) -> bool:
    if DEFUSE:
        return False
    return bool(getattr(m, "valid_for_acceleration", False)) and not getattr(m, "bos", False) \
        and getattr(m, "split", 0) == 0


def _qp(m):
    return (m.act_scales_inv, m.act_zero_points)


def _same_qparams(a, b) -> bool:
    """Host comparison of two layers' activation qparams (a device sync: cached by callers, and
    only ever evaluated in the eager warm-up that precedes graph capture)."""
    return bool(torch.equal(a.act_scales_inv, b.act_scales_inv)
                and torch.equal(a.act_zero_points, b.act_zero_points))


def _fusable_f16(x) -> bool:
    return x.is_cuda and x.dtype == torch.float16


# ---------------------------------------------------------------------------------------------
# Derived state of the fused graph (packed GEMM operands, quantizer groupings) is cached on the
# modules.  Two rules keep the caches honest:
#   * a cached DECISION (do these layers share one activation quantizer?) is valid only for the
#     very tensors it was taken on, at their current in-place `_version` -- module swaps
#     (quantize_unet after an FP16 run), `.to()` and load_state_dict / broadcast all invalidate it;
#   * cached DATA never goes stale because it is not a copy: the row-concatenated tensors of a
#     packed GEMM become the storage and the layers' own buffers become views of them, so in-place
#     updates reach the packed launch -- and every captured graph -- by construction.
# ---------------------------------------------------------------------------------------------
def _quantizer_groups(holder: dict, key, layers):
    """Group id per layer: equal activation qparams <=> equal id.  One device->host copy (a sync:
    only ever taken in the eager warm-up that precedes graph capture), memoised in `holder[key]`
    against the identity and in-place version of the qparam tensors."""
    src = [t for m in layers for t in (m.act_scales_inv, m.act_zero_points)]
    ver = tuple(t._version for t in src)
    e = holder.get(key)
    if (e is not None and len(e[0]) == len(src) and all(a is b for a, b in zip(e[0], src))
            and e[1] == ver):
        return e[2]
    ids = []
    if layers:
        vals = torch.stack([t.detach().reshape(-1)[0].float() for t in src]).cpu().tolist()
        seen = {}
        for i in range(len(layers)):
            ids.append(seen.setdefault((vals[2 * i], vals[2 * i + 1]), len(seen)))
    holder[key] = (src, ver, ids)
    return ids


def _memo(mod) -> dict:
    return mod.__dict__.setdefault("_mixdq_memo", {})


_PACK_VECS = (("wscale", "weight_scales"), ("wsum", "weight_sum_by_input_channels"),
              ("scale", "scale"), ("bias0", "bias0"))


def _uniform_storage(layers) -> bool:
    """Layers that are to share one GEMM must store their weights alike (all int8 or all packed
    4-bit).  A pure check: the forward never changes a module's buffers -- mixed groups are
    unified once, by `unify_packed_storage_` from `SDXLUNet.set_fused`, or run unpacked."""
    return len({bool(m.w_packed4) for m in layers}) == 1


def unify_packed_storage_(layers) -> bool:
    """Explicit post-conversion step (SDXLUNet.set_fused -> prepare_fused_): in a mixed-precision
    config to_q / to_k / to_v (or to_k / to_v) may be a mix of int8 and packed 4-bit layers; the
    packed ones are widened to int8 STORAGE (the values stay the 4-bit integers, so every output
    is unchanged) -- one launch instead of three is worth more than those layers' halved bytes.
    Replaces the `weight_int4` buffer by `weight_int`: state-dict keys change HERE, once, on every
    rank alike, never inside a forward.  Returns whether the group is uniform afterwards."""
    if _uniform_storage(layers):
        return True
    if any(getattr(m, "weight_int4", None) is None and m.w_packed4 for m in layers):
        return False
    with torch.no_grad():
        for m in layers:
            if m.w_packed4:
                w = m._weight_values().contiguous()
                del m.weight_int4
                m.register_buffer("weight_int", w)
                m.w_packed4 = False
    return True


def _pack_rows(layers):
    """Row-concatenate the weights and per-channel epilogue vectors of Linear layers that read the
    same INT8 operand: one GEMM against [sum N_i, K] computes every output element exactly as the
    separate launches do.  The concatenated tensors are the storage; the layers keep views."""
    w4 = bool(layers[0].w_packed4)
    names = (("w", "weight_int4" if w4 else "weight_int"),) + _PACK_VECS
    pack = dict(w4=w4, C=layers[0].out_features, layers=tuple(weakref.ref(m) for m in layers),
                names=names)
    with torch.no_grad():
        for key, name in names:
            cat = torch.cat([getattr(m, name) for m in layers], dim=0).contiguous()
            off = 0
            for m in layers:
                n = getattr(m, name).shape[0]
                setattr(m, name, cat[off:off + n])        # registered buffer -> view of the pack
                off += n
            pack[key] = cat
    return pack


def _pack_valid(pack, layers) -> bool:
    """The layers' buffers still alias the pack (false after a module swap, `.to()`, deepcopy...)."""
    if pack is None or len(pack["layers"]) != len(layers) or any(
            a() is not b for a, b in zip(pack["layers"], layers)):
        return False
    for key, name in pack["names"]:
        cat, off = pack[key], 0
        row = cat.stride(0) * cat.element_size()
        for m in layers:
            t = getattr(m, name, None)
            if (not torch.is_tensor(t) or t.device != cat.device
                    or t.data_ptr() != cat.data_ptr() + off * row):
                return False
            off += t.shape[0]
    return True


def _refresh_after_load(mod, _incompatible_keys):
    mod.refresh_derived_()


def _gn_feed(norm: nn.GroupNorm, x, consumer, silu: bool, x2=None, raw_for=None):
    """GroupNorm(+SiLU) of a channels-last fp16 x for `consumer`: returns (tensor, quantized?).
    x2: the norm is over cat([x, x2], dim=1), read from the two tensors in place.
    raw_for: a W8A8 conv that reads the norm's INPUT (the ResNet 1x1 shortcut, split when x2 is
    given); the return value gains a third element: the input(s) quantized for it by the same pass
    (a list, one INT8 tensor per source), or None when that pass is not the HIP kernel."""
    from mixdq_amd import _C
    if DEFUSE and x2 is not None:
        x, x2 = torch.cat([x, x2], dim=1), None
    N, C, H, W = x.shape
    C += 0 if x2 is None else x2.shape[1]
    if (_fusable_f16(x) and x.is_contiguous(memory_format=torch.channels_last)
            and (x2 is None or (_fusable_f16(x2)
                                and x2.is_contiguous(memory_format=torch.channels_last)
                                and x.shape[1] % 8 == 0 and x2.shape[1] % 8 == 0))
            and _C.groupnorm_supported(N, H * W, C, norm.num_groups)):
        raw_qp = None
        if (GN_RAW_OUTPUTS and not DEFUSE and raw_for is not None
                and getattr(raw_for, "valid_for_acceleration", False)
                and not getattr(raw_for, "bos", False)
                and raw_for.split == (0 if x2 is None else x.shape[1])):
            raw_qp = [(raw_for.act_scales_inv, raw_for.act_zero_points)]
            if x2 is not None:
                raw_qp.append((raw_for.act_scales_inv_0, raw_for.act_zero_points_0))
        if _accel(consumer):
            out = _C.groupnorm_silu_quantize(x, norm.num_groups, norm.weight, norm.bias, norm.eps,
                                             *_qp(consumer), silu=silu, x2=x2, raw_qparams=raw_qp)
            res = (out[0], True)
        else:
            out = _C.groupnorm_silu_quantize(x, norm.num_groups, norm.weight, norm.bias, norm.eps,
                                             silu=silu, want_f16=True, x2=x2, raw_qparams=raw_qp)
            res = (out[1], False)
        if raw_for is None:
            return res
        return res + (out[2] if raw_qp is not None else None,)
    if x2 is not None:
        x = torch.cat([x, x2], dim=1)
    # (the stock op by FUNCTION: a module swapped by swap_glue may fold the SiLU into its own launch)
    h = F.group_norm(x, norm.num_groups, norm.weight, norm.bias, norm.eps)
    res = ((F.silu(h) if silu else h), False)
    return res if raw_for is None else res + (None,)


_SHORTCUT_STREAMS = {}
# ResNet 1x1 shortcut convs beside norm1 / conv1 / norm2 on a side stream?  Off: measured on
# MI355X / ROCm 7.2, every extra branch in the captured hipGraph costs more than the kernels it
# overlaps (shortcut branch: 14.40 -> 13.70 ms per step without it; the K/V + time-embedding
# branch: -0.5 ms) -- the step is one chain of launches on one stream.  MIXDQ_SHORTCUT_STREAM=1
# restores the branch for A/B runs.
SHORTCUT_SIDE_STREAM = __import__("os").environ.get("MIXDQ_SHORTCUT_STREAM", "0") == "1"
# norm1's apply pass also writes the shortcut conv's INT8 operand(s) (MIXDQ_GN_RAW=0: separate
# quantize launches, for A/B runs)
GN_RAW_OUTPUTS = __import__("os").environ.get("MIXDQ_GN_RAW", "1") == "1"


def _shortcut_stream(device):
    s = _SHORTCUT_STREAMS.get(device)
    if s is None:
        s = _SHORTCUT_STREAMS[device] = torch.cuda.Stream(device=device)
    return s


def _ln_feed(norm: nn.LayerNorm, x, consumers):
    """LayerNorm of x [B, T, C] for several consumer layers: one fused kernel produces an int8
    tensor per distinct activation quantizer (and the fp16 tensor if some consumer needs it).
    Returns a list of (tensor, quantized?) aligned with `consumers`."""
    from mixdq_amd import _C
    C = x.shape[-1]
    if not (_fusable_f16(x) and x.is_contiguous() and C % 16 == 0 and C <= 2048):
        h = F.layer_norm(x, norm.normalized_shape, norm.weight, norm.bias, norm.eps)
        return [(h, False)] * len(consumers)
    groups, slot = _ln_plan(norm, consumers)
    want_f16 = any(s < 0 for s in slot)
    outs, h = _C.layernorm_quantize(x, norm.weight, norm.bias, norm.eps, groups, want_f16=want_f16)
    return [((outs[s], True) if s >= 0 else (h, False)) for s in slot]


def _ln_plan(norm, consumers):
    """(distinct (scale_inv, zero_point) pairs among the accelerated consumers, slot of each consumer in that
    list or -1 for one that takes the FP16 tensor) -- cached on the norm, re-derived when the layers change."""
    acc = [c for c in consumers if _accel(c)]
    ids = _quantizer_groups(_memo(norm), "ln", acc)
    plan = norm.__dict__.get("_mixdq_plan")
    if (plan is None or len(plan[0]) != len(consumers)
            or any(a() is not b for a, b in zip(plan[0], consumers)) or plan[3] is not ids):
        groups, slot, it = {}, [], iter(ids)   # distinct quantizers among the accelerated consumers
        for c in consumers:
            if not _accel(c):
                slot.append(-1)
                continue
            gi = next(it)
            groups.setdefault(gi, c)
            slot.append(gi)
        # weak references: a plan must not keep swapped-out float layers (and their FP16
        # weights) alive until the next forward
        plan = norm.__dict__["_mixdq_plan"] = (tuple(weakref.ref(c) for c in consumers),
                                               [_qp(groups[g]) for g in sorted(groups)], slot, ids)
    return plan[1], plan[2]


# the largest problem a GEMM + LayerNorm launch takes (csrc/igemm_ln.hip select_ln: one 64 x 80 or 128 x 80 tile
# per CU): M <= 4096 at N = 640, M <= 2048 at N = 1280 -- the workspace is sized for it ONCE
_LN_WS_ROWS = 4096


def _ln_workspace(owner, M, N, device):
    """The exchange buffer of the GEMM + LayerNorm launches of one network: kept on `owner` (the SDXLUNet, whose
    launches run one at a time on one stream), per device, allocated ONCE for the largest problem the launch form
    takes (2.1 MB) and never replaced -- a hipGraph captured through hip_graph_opt holds its raw pointer, so a
    buffer that grew by reallocation would leave every earlier graph writing records and counters into memory the
    allocator has handed to someone else.  The epoch in its first page only grows (launch tags); the
    departure counter is zero between launches; the third word is the sticky error word (_C.qlinear_ln_status)."""
    from mixdq_amd import _C
    store = owner.__dict__.setdefault("_ln_ws", {})
    ws = store.get(device)
    if ws is None:
        ws = store[device] = _C.qlinear_ln_workspace(max(int(M), _LN_WS_ROWS), max(int(N), 1280), device)
    need = int(_C._lib.mixdq_qlinear_ln_workspace_bytes(int(M), int(N)))
    if ws.numel() < need:      # (cannot happen for a shape the launch form takes; never silently reallocate)
        raise RuntimeError(f"GEMM + LayerNorm workspace of {ws.numel()} bytes is too small for M={M}, N={N}")
    return ws


def _gemm_res_ln(layer, x_int, residual, next_ln):
    """residual + layer(x_int) for an already quantized operand, AND -- when `next_ln` = (norm, consumers,
    owner) names the LayerNorm that reads the result -- that LayerNorm's feeds (as `_ln_feed` returns them):
    ONE launch where the GEMM + LayerNorm kernel takes the shape (mixdq_qlinear_w8a8_ln: the column tiles of
    a row block exchange their partial statistics), else the GEMM followed by the LayerNorm launch.  The
    bits are the same either way.  Returns (y, feeds or None)."""
    from mixdq_amd import _C
    if next_ln is None:
        return layer.forward_quantized(x_int, residual=residual), None
    norm, consumers, owner = next_ln
    N, K = layer.out_features, layer.in_features
    M = x_int.numel() // K
    if (not DEFUSE and _accel(layer) and not layer.w_packed4 and x_int.dtype == torch.int8
            and (residual is None or residual.is_contiguous())
            and norm.weight.dtype == torch.float16 and _C.qlinear_ln_supported(M, N, K)):
        groups, slot = _ln_plan(norm, consumers)
        want_f16 = any(s_ < 0 for s_ in slot)
        y, outs, h = _C.qlinear_ln(x_int, layer.weight_int, layer.scale, layer.bias0, layer.bias, residual,
                                   norm.weight, norm.bias, norm.eps, groups,
                                   _ln_workspace(owner, M, N, x_int.device), want_f16=want_f16)
        return y, [((outs[s_], True) if s_ >= 0 else (h, False)) for s_ in slot]
    y = layer.forward_quantized(x_int, residual=residual)
    return y, _ln_feed(norm, y, consumers)


def _run(layer, feed, residual=None):
    """Run a Linear on a (tensor, quantized?) feed; fold `residual` into the epilogue if possible."""
    t, quantized = feed
    if quantized:
        return layer.forward_quantized(t, residual=residual)
    if residual is not None and _fp_layer(layer) and not DEFUSE:
        return layer.forward_fp(t, residual=residual)       # FP16 kernel, add in its epilogue
    y = layer(t)
    return y if residual is None else y + residual


def _fp_layer(layer) -> bool:
    """A swapped layer on the reference's FP fallback (own FP16 kernels, residual-capable)."""
    return hasattr(layer, "forward_fp") and not getattr(layer, "valid_for_acceleration", True)


def _linear_res(layer, x, residual):
    """layer(x) + residual with the add folded into the GEMM epilogue when the layer is W8A8."""
    if _accel(layer) and _fusable_f16(x) and residual.is_contiguous():
        from mixdq_amd.nn.Linear import quant_op
        return layer.forward_quantized(quant_op(x, *_qp(layer)), residual=residual)
    if _fp_layer(layer) and _fusable_f16(x) and not DEFUSE:
        return layer.forward_fp(x, residual=residual)
    return layer(x) + residual


SDXL_CONFIG = dict(
    in_channels=4, out_channels=4,
    block_out_channels=(320, 640, 1280),
    layers_per_block=2,
    transformer_layers_per_block=(0, 2, 10),   # 0 = plain DownBlock2D / UpBlock2D
    mid_transformer_layers=10,
    head_dim=64,
    cross_attention_dim=2048,
    time_embed_dim=1280,
    addition_time_embed_dim=256,
    projection_class_embeddings_input_dim=2816,
    norm_num_groups=32,
)


def sinusoidal_embedding(t: torch.Tensor, dim: int) -> torch.Tensor:
    """diffusers Timesteps(dim, flip_sin_to_cos=True, downscale_freq_shift=0)."""
    half = dim // 2
    key = (half, t.device)
    freqs = _SIN_FREQS.get(key)
    if freqs is None:      # a constant of (dim, device): four launches per call when recomputed
        exponent = -math.log(10000.0) * torch.arange(half, dtype=torch.float32, device=t.device) / half
        freqs = torch.exp(exponent)
        if not (t.is_cuda and torch.cuda.is_current_stream_capturing()):
            _SIN_FREQS[key] = freqs        # (never cache a tensor that only a graph replay fills)
    emb = t.float()[:, None] * freqs[None, :]
    return torch.cat([torch.cos(emb), torch.sin(emb)], dim=-1)


_SIN_FREQS = {}


class TimestepEmbedding(nn.Module):
    def __init__(self, in_dim, dim):
        super().__init__()
        self.linear_1 = nn.Linear(in_dim, dim)
        self.linear_2 = nn.Linear(dim, dim)

    def forward(self, x):
        return self.linear_2(F.silu(self.linear_1(x)))


class ResnetBlock2D(nn.Module):
    def __init__(self, cin, cout, temb_dim, groups, split=0):
        super().__init__()
        self.norm1 = nn.GroupNorm(groups, cin, eps=1e-5)
        self.conv1 = nn.Conv2d(cin, cout, 3, 1, 1)
        self.time_emb_proj = nn.Linear(temb_dim, cout)
        self.norm2 = nn.GroupNorm(groups, cout, eps=1e-5)
        self.conv2 = nn.Conv2d(cout, cout, 3, 1, 1)
        self.conv_shortcut = nn.Conv2d(cin, cout, 1, 1, 0) if cin != cout else None
        self.nonlinearity = nn.SiLU()      # a MODULE, as in diffusers' ResnetBlock2D (swappable by type: nn/glue.py)
        if self.conv_shortcut is not None and split:
            # up-block: input = cat(hidden[:split], skip); the quantized shortcut uses separate
            # activation scales for the two halves (quant_block_forward_func.py:153-157)
            self.conv_shortcut.split = split

    fused = False

    def forward(self, x, temb, skip=None):
        """skip: the up-block's skip connection; the input is cat([x, skip], dim=1).  The fused
        path reads the two tensors in place (two-source GroupNorm, split shortcut) when it can."""
        if self.fused and _fusable_f16(x):
            if skip is not None and not self._pair_ok(x, skip):
                x, skip = torch.cat([x, skip], dim=1), None
            return self.forward_fused(x, temb, skip)
        if skip is not None:
            x = torch.cat([x, skip], dim=1)
        h = self.conv1(self.nonlinearity(self.norm1(x)))
        h = h + self.time_emb_proj(self.nonlinearity(temb))[:, :, None, None]
        h = self.conv2(self.nonlinearity(self.norm2(h)))
        if self.conv_shortcut is not None:
            x = self.conv_shortcut(x)
        return x + h

    def _pair_ok(self, x, skip) -> bool:
        """The (hidden, skip) pair can stay unconcatenated: the shortcut is a W8A8 split layer whose
        split is exactly `hidden`, and both tensors are channels-last with channel counts % 8."""
        sc = self.conv_shortcut
        return bool(not DEFUSE and sc is not None and getattr(sc, "valid_for_acceleration", False)
                    and getattr(sc, "split", 0) == x.shape[1] and _fusable_f16(skip)
                    and x.is_contiguous(memory_format=torch.channels_last)
                    and skip.is_contiguous(memory_format=torch.channels_last)
                    and x.shape[1] % 8 == 0 and skip.shape[1] % 8 == 0)

    def forward_fused(self, x, temb, skip=None):
        ahead = self.__dict__.pop("_t", None)
        if ahead is not None:       # projected ahead of time (SDXLUNet._project_temb_ahead)
            t, ready = ahead
            if ready is not None:
                torch.cuda.current_stream().wait_event(ready)
                t.record_stream(torch.cuda.current_stream())
        else:
            t = self.time_emb_proj(F.silu(temb))                   # [N, Cout]
        sc = None
        side = None
        if self.conv_shortcut is not None and SHORTCUT_SIDE_STREAM and skip is None:
            # the 1x1 shortcut (quantize + GEMM) only meets the main path at conv2's residual add:
            # beside norm1 / conv1 / norm2 on a side stream (A/B runs only, see above)
            main, side = torch.cuda.current_stream(), _shortcut_stream(x.device)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                sc = self.conv_shortcut(x)
            x.record_stream(side)
            feed, q = _gn_feed(self.norm1, x, self.conv1, silu=True)
        elif self.conv_shortcut is not None:
            # norm1's apply pass holds the block input in registers: it also writes the shortcut's
            # INT8 operand(s) -- each half of a split shortcut with its own quantizer, where it lies
            # (no concatenation) -- instead of separate quantize launches
            feed, q, raw = _gn_feed(self.norm1, x, self.conv1, silu=True, x2=skip,
                                    raw_for=self.conv_shortcut)
            if raw is not None and skip is not None:
                sc = self.conv_shortcut.forward_parts_quantized(raw[0], raw[1])
            elif raw is not None:
                sc = self.conv_shortcut.forward_quantized(raw[0])
            elif skip is not None:
                sc = self.conv_shortcut.forward_parts(x, skip)
            else:
                sc = self.conv_shortcut(x)
        else:
            feed, q = _gn_feed(self.norm1, x, self.conv1, silu=True, x2=skip)
        if q:   # h = conv1(..) + t[:, :, None, None], the add folded into the conv epilogue
            h = self.conv1.forward_quantized(feed, residual=t.contiguous(), residual_per_image=True)
        elif _fp_layer(self.conv1) and not DEFUSE:
            h = self.conv1.forward_fp(feed, residual=t.contiguous(), residual_per_image=True)
        else:
            h = self.conv1(feed) + t[:, :, None, None]
        if sc is not None:
            if side is not None:
                main.wait_stream(side)
                sc.record_stream(main)
            x = sc
        feed, q = _gn_feed(self.norm2, h, self.conv2, silu=True)
        if q and x.is_contiguous(memory_format=torch.channels_last):
            return self.conv2.forward_quantized(feed, residual=x)  # x + conv2(..)
        if not q and _fp_layer(self.conv2) and not DEFUSE:
            return self.conv2.forward_fp(feed, residual=x)
        return x + self.conv2(feed)


class Attention(nn.Module):
    def __init__(self, dim, cross_dim, head_dim):
        super().__init__()
        self.heads = dim // head_dim
        kv_dim = cross_dim if cross_dim is not None else dim
        self.to_q = nn.Linear(dim, dim, bias=False)
        self.to_k = nn.Linear(kv_dim, dim, bias=False)
        self.to_v = nn.Linear(kv_dim, dim, bias=False)
        self.to_out = nn.ModuleList([nn.Linear(dim, dim), nn.Identity()])

    def forward(self, x, context=None):
        context = x if context is None else context
        return self.to_out[0](self.attend(self.to_q(x), self.to_k(context), self.to_v(context)))

    def attend(self, q, k, v):
        B, T, C = q.shape
        h = self.heads
        q = q.unflatten(-1, (h, C // h)).transpose(1, 2)    # views, also for column slices of a
        k = k.unflatten(-1, (h, C // h)).transpose(1, 2)    # fused q|k|v projection
        v = v.unflatten(-1, (h, C // h)).transpose(1, 2)
        o = F.scaled_dot_product_attention(q, k, v)   # FP16, as in the reference
        return o.transpose(1, 2).reshape(B, T, C)

    def cross_attend_out(self, feed_q, k, v, residual, next_ln=None):
        """residual + to_out[0](attention(to_q(feed_q), k, v)) for the cross-attention, whose keys /
        values (77 text tokens) fit one workgroup's LDS: to_q's INT8 GEMM, the attention core and
        to_out.0's quantizer are ONE launch; q never exists in memory.  Bit-identical to the
        three-step path it replaces (tests/test_fused_gpu.py)."""
        if _cross_fusable(self, feed_q, k, v, residual):
            from mixdq_amd import _C
            q, out = self.to_q, self.to_out[0]
            o_int = _C.qlinear_attention(feed_q[0], q.weight_int4 if q.w_packed4 else q.weight_int,
                                         q.scale, q.bias0, k, v, *_qp(out), _w4=q.w_packed4)
            return _gemm_res_ln(out, o_int, residual, next_ln)
        return self.attend_out(_run(self.to_q, feed_q), k, v, residual, next_ln=next_ln)

    def attend_out(self, q, k, v, residual, prefetch=None, next_ln=None):
        """residual + to_out[0](attention(q, k, v)) on the fused path: the HIP FP16 attention core
        reads q/k/v in place (column slices of the fused projection included) and, when to_out[0]
        is W8A8, emits its INT8 operand directly; the residual add rides in the GEMM epilogue."""
        from mixdq_amd import _C
        out = self.to_out[0]
        C = q.shape[-1]
        ok = (C == self.heads * 64 and all(_fusable_f16(t) and t.dim() == 3 and t.stride(-1) == 1
                                           and t.stride(0) % 8 == 0 and t.stride(1) % 8 == 0
                                           and t.data_ptr() % 16 == 0 for t in (q, k, v)))
        def plain(y):          # (y, feeds of the LayerNorm that reads y) by two launches
            return y, (None if next_ln is None else _ln_feed(next_ln[0], y, next_ln[1]))
        if not ok:
            return plain(_linear_res(out, self.attend(q, k, v), residual))
        if _accel(out) and residual.is_contiguous():
            o_int = _C.attention_f16(q, k, v, self.heads, *_qp(out), _prefetch=prefetch)
            return _gemm_res_ln(out, o_int, residual, next_ln)
        return plain(_linear_res(out, _C.attention_f16(q, k, v, self.heads, _prefetch=prefetch), residual))


CROSS_FUSE_MAX_ROWS = int(__import__("os").environ.get("MIXDQ_CROSS_FUSE_MAX_ROWS", "4096"))
# LayerNorms in the epilogue of the GEMM that produces their input (csrc/igemm_ln.hip).  Off by default: the step
# time is EQUAL either way (10.975 / 10.986 ms with, 10.965 / 10.945 without, same box: DESIGN.md section 3.13) -- 151
# launches fewer, paid for by a cross-workgroup exchange per launch -- and the plain form keeps every launch free of
# inter-workgroup waits.  MIXDQ_LN_CHAIN=1 (or `mixdq_amd.unet.LN_CHAIN = True`) turns it on; bench.py times both.
LN_CHAIN = __import__("os").environ.get("MIXDQ_LN_CHAIN", "0") == "1"
# weight prefetch from the self-attention launch: on for launches of up to this many rows
PREFETCH = __import__("os").environ.get("MIXDQ_PREFETCH", "1") != "0"
WEIGHT_ARENA = __import__("os").environ.get("MIXDQ_WEIGHT_ARENA", "0") == "1"      # arena.py (prepare_fused_)
PREFETCH_MAX_ROWS = int(__import__("os").environ.get("MIXDQ_PREFETCH_MAX_ROWS", "0"))      # 0: no limit (see _build_prefetch_plan)
PREFETCH_MB_PER_LAUNCH = float(__import__("os").environ.get("MIXDQ_PREFETCH_MB", "48"))   # per 1024 x 1024 scores
PREFETCH_MAX_LEAD = int(__import__("os").environ.get("MIXDQ_PREFETCH_LEAD", "4"))         # launches a weight may be read ahead
PREFETCH_SKIP_MB = float(__import__("os").environ.get("MIXDQ_PREFETCH_SKIP_MB", "0"))       # experiment: leave tensors above this cold (0: off)


def _trace_signature(trace):
    return tuple(it if isinstance(it, tuple) else (it.data_ptr(), it.numel()) for it in trace)


def _plan_alive(plan, device) -> bool:
    """Every tensor of the plan still exists, at the address it was planned at, on `device`."""
    for lst in plan["lists"]:
        for ref, ptr in zip(lst, plan["ptrs"][id(lst)]):
            t = ref()
            if t is None or t.device != device or t.data_ptr() != ptr:
                return False
    return True


def _build_prefetch_plan(trace):
    """Which weights does each self-attention launch of a forward read ahead (mixdq_attention_f16_prefetch)?
    `trace` (mixdq_amd._C.PrefetchContext.trace) is the forward's execution order: weight operands as the GEMM / conv entry
    points saw them, and a marker (rows, keys) per long-key attention launch.  Launch j is given the weights
    used between it and launch j + 1 -- the rest of its transformer block, the next block's q|k|v; at the
    end of a Transformer2DModel also proj_out, the ResNet convs, shortcuts and samplers up to the next
    transformer -- up to a byte budget proportional to the launch's own work (it should not outlast the
    attention) and 16 ranges; what does not fit is offered to launch j - 1, ... j - PREFETCH_MAX_LEAD behind
    THEIR own intervals (read earlier still: the 256 MB Infinity Cache keeps a few hundred microseconds of
    the step's weight stream, not more -- an unbounded cascade measured slower than no look-back at all),
    and stays cold if none of them has room.  MIXDQ_PREFETCH_MAX_ROWS (default 0: off) leaves launches of more
    rows without a payload: from batch 2 on the attention launch fills the chip, so its payload workgroups run in
    the launch's tail and lengthen it -- measured, the weights they leave in the cache are still worth more
    (batch 8: 48.65 -> 47.98 ms, batch 2: 17.17 -> 16.98, batch 4: 27.73 -> 27.60; 4096 was the first default).
    Weights used before the first attention launch have no host and stay cold."""
    marks = [(i, it) for i, it in enumerate(trace) if isinstance(it, tuple)]
    if not marks:
        return None
    lists = [[] for _ in marks]
    carry = []                                       # (tensor, launches it has been moved back)
    nbytes = lambda t: t.numel() * t.element_size()  # noqa: E731
    for j in range(len(marks) - 1, -1, -1):
        lo = marks[j][0] + 1
        hi = marks[j + 1][0] if j + 1 < len(marks) else len(trace)
        own, seen = [], set()
        for t in trace[lo:hi]:
            if isinstance(t, tuple) or t.data_ptr() in seen or nbytes(t) < (64 << 10):
                continue
            if PREFETCH_SKIP_MB and nbytes(t) > PREFETCH_SKIP_MB * 1e6:
                continue
            seen.add(t.data_ptr())
            own.append(t)
        _, rows, keys = marks[j][1]
        budget = (0 if PREFETCH_MAX_ROWS and rows > PREFETCH_MAX_ROWS
                  else PREFETCH_MB_PER_LAUNCH * 1e6 * rows * keys / float(1 << 20))
        budget = min(budget, 160e6)
        new_carry, blocked = [], False
        for t in own:                                # the launch's own interval first, in order
            if not blocked and nbytes(t) <= budget and len(lists[j]) < 16:
                lists[j].append(t)
                budget -= nbytes(t)
            else:
                blocked = True
                new_carry.append((t, 1))
        for t, d in carry:                           # then what later launches could not take
            if t.data_ptr() in seen:
                continue
            seen.add(t.data_ptr())
            if nbytes(t) <= budget and len(lists[j]) < 16:
                lists[j].append(t)
                budget -= nbytes(t)
            elif d < PREFETCH_MAX_LEAD:
                new_carry.append((t, d + 1))         # else: stays cold
        carry = new_carry
    import weakref
    ref_lists = [[weakref.ref(t) for t in lst] for lst in lists]       # never keeps a replaced weight alive
    ptrs = {id(rl): [t.data_ptr() for t in lst] for rl, lst in zip(ref_lists, lists)}
    return {"lists": ref_lists, "ptrs": ptrs, "sig": _trace_signature(trace)}


def _cross_fusable(attn, feed, k, v, residual) -> bool:
    """to_q + cross-attention + quantize for to_out.0 as ONE launch (mixdq_qlinear_w8a8_attn)?"""
    from mixdq_amd import _C
    q, out = attn.to_q, attn.to_out[0]
    t, quantized = feed
    return bool(not DEFUSE and quantized and _accel(q) and q.bias is None and _accel(out) and t.dim() == 3
                and attn.heads * 64 == q.out_features and residual.is_contiguous()
                and all(_fusable_f16(z) and z.dim() == 3 and z.stride(-1) == 1
                        and z.stride(0) % 8 == 0 and z.stride(1) % 8 == 0
                        and z.data_ptr() % 16 == 0 for z in (k, v))
                and k.shape == v.shape and k.shape[0] == t.shape[0]
                # the fused launch runs the GEMM on 64 x 128 tiles: from a few thousand rows on, the
                # large-tile GEMM + the attention kernel are faster (batch 8: 26 + 20 us vs 56 us)
                and t.shape[0] * t.shape[1] <= CROSS_FUSE_MAX_ROWS
                and _C.qlinear_attention_supported(t.shape, q.out_features, q.in_features, k))


class GEGLU(nn.Module):
    def __init__(self, dim, inner):
        super().__init__()
        self.proj = nn.Linear(dim, inner * 2)

    def forward(self, x):
        x, gate = self.proj(x).chunk(2, dim=-1)
        return x * F.gelu(gate)


class FeedForward(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.net = nn.ModuleList([GEGLU(dim, dim * 4), nn.Identity(), nn.Linear(dim * 4, dim)])

    def forward(self, x):
        assert not self.__dict__.get("_interleaved"), "rows are interleaved: fused path only"
        return self.net[2](self.net[0](x))

    def set_interleaved(self, on: bool):
        """Store ff.net.0.proj's rows as value|gate groups of 16 (on) or in the ordinary
        [values | gates] order (off).  Interleaved, one GEMM launch produces net.2's INT8 operand
        (GEMM + GEGLU + quantize); only W8A8 pairs qualify."""
        from mixdq_amd import _C
        proj, out_layer = self.net[0].proj, self.net[2]
        have = bool(self.__dict__.get("_interleaved"))
        if on == have:
            return
        if on and not (_accel(proj) and _accel(out_layer) and proj.out_features % 64 == 0
                       and proj.in_features % 16 == 0 and proj.weight_scales.is_cuda):
            return
        perm = _C.geglu_row_order(proj.out_features // 2, proj.weight_scales.device)
        if not on:
            perm = torch.argsort(perm)
        proj.permute_output_rows_(perm)
        self.__dict__["_interleaved"] = on

    def forward_fused(self, feed, residual, next_ln=None):
        """residual + net2(geglu(proj(feed))): GEGLU fused with net.2's quantizer, the residual add
        folded into net.2's epilogue.  With `next_ln` (the LayerNorm that reads the result: the next
        block's norm1) returns (y, that LayerNorm's feeds), the LayerNorm riding in net.2's launch where
        the GEMM + LayerNorm kernel takes the shape (`_gemm_res_ln`)."""
        from mixdq_amd import _C
        out_layer = self.net[2]
        if next_ln is not None:
            y = self.forward_fused(feed, residual) if (DEFUSE or not self.__dict__.get("_interleaved")) else None
            if y is not None:
                return y, _ln_feed(next_ln[0], y, next_ln[1])
            assert feed[1], "interleaved rows need the quantized feed"
            q = self.net[0].proj.forward_quantized_geglu(feed[0], out_layer)
            return _gemm_res_ln(out_layer, q, residual, next_ln)
        if self.__dict__.get("_interleaved") and not DEFUSE:
            assert feed[1], "interleaved rows need the quantized feed"
            q = self.net[0].proj.forward_quantized_geglu(feed[0], out_layer)
            return out_layer.forward_quantized(q, residual=residual)
        h = _run(self.net[0].proj, feed)
        if self.__dict__.get("_interleaved"):     # DEFUSE: columns back to [values | gates]
            D2 = h.shape[-1]
            inv = torch.argsort(_C.geglu_row_order(D2 // 2, h.device))
            h = h.index_select(-1, inv)
        D = h.shape[-1] // 2
        if _fusable_f16(h) and h.is_contiguous() and D % 8 == 0:
            if _accel(out_layer):
                q, _ = _C.geglu_quantize(h, *_qp(out_layer))
                return out_layer.forward_quantized(q, residual=residual)
            _, g = _C.geglu_quantize(h, want_f16=True)
            return _run(out_layer, (g, False), residual)
        a, gate = h.chunk(2, dim=-1)
        return out_layer(a * F.gelu(gate)) + residual


class BasicTransformerBlock(nn.Module):
    def __init__(self, dim, cross_dim, head_dim):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim)
        self.attn1 = Attention(dim, None, head_dim)
        self.norm2 = nn.LayerNorm(dim)
        self.attn2 = Attention(dim, cross_dim, head_dim)
        self.norm3 = nn.LayerNorm(dim)
        self.ff = FeedForward(dim)

    fused = False

    def forward(self, x, context):
        if self.fused and _fusable_f16(x):
            return self.forward_fused(x, context)
        x = x + self.attn1(self.norm1(x))
        x = x + self.attn2(self.norm2(x), context)
        return x + self.ff(self.norm3(x))

    def _qkv_fused(self):
        """Self-attention to_q / to_k / to_v read the same normalised tensor; when all three are
        W8A8 with the SAME activation quantizer (they are calibrated on the same data) and carry no
        bias, one GEMM against the row-concatenated weights [3C, C] replaces three: per-channel
        scale / bias0 simply concatenate, so every output element is computed exactly as before
        (`_pack_rows`).  Nothing negative is cached: the decision is re-taken whenever the layers
        or their quantizer tensors change (e.g. quantize_unet after an FP16 run of this graph)."""
        a = self.attn1
        layers = [a.to_q, a.to_k, a.to_v]
        if DEFUSE or not all(_accel(m) and m.bias is None for m in layers):
            return None
        if len(set(_quantizer_groups(_memo(self), "qkv", layers))) != 1:
            return None
        if not _uniform_storage(layers):
            return None
        pack = self.__dict__.get("_qkv")
        if not _pack_valid(pack, layers):
            pack = self.__dict__["_qkv"] = _pack_rows(layers)
        return pack

    def ln1_consumers(self):
        a = self.attn1
        return [a.to_q, a.to_k, a.to_v]

    def forward_fused(self, x, context, feeds1=None, next_ln=None, owner=None):
        """`feeds1`: norm1's feeds when the GEMM that produced x already computed them (`_gemm_res_ln`);
        `next_ln`: the LayerNorm that reads this block's output (the next block's norm1) -- then returns
        (x, its feeds); `owner`: the module that keeps the GEMM + LayerNorm exchange buffer (None: the
        LayerNorms of this block run as launches of their own)."""
        x = x.contiguous()
        a = self.attn1
        pack = self._qkv_fused()
        feeds = feeds1 if feeds1 is not None else _ln_feed(self.norm1, x, self.ln1_consumers())
        ln2 = None if owner is None else (self.norm2, [self.attn2.to_q], owner)
        ln3 = None if owner is None else (self.norm3, [self.ff.net[0].proj], owner)
        if pack is not None and feeds[0][1]:
            from mixdq_amd.op.qlinear import qlinear
            q0 = a.to_q
            qkv = qlinear(feeds[0][0], pack["w"], pack["wscale"], q0.act_scales, q0.act_zero_points,
                          pack["wsum"], pack["scale"], pack["bias0"], None, _w4=pack["w4"])
            C = pack["C"]
            x, f2 = a.attend_out(qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:], x, next_ln=ln2)
        else:
            fq, fk, fv = feeds                                      # x + attn1(norm1(x))
            x, f2 = a.attend_out(_run(a.to_q, fq), _run(a.to_k, fk), _run(a.to_v, fv), x, next_ln=ln2)
        a = self.attn2
        (fq,) = f2 if f2 is not None else _ln_feed(self.norm2, x, [a.to_q])
        kv = self.__dict__.pop("_kv", None)
        if kv is not None:          # projected ahead of time (SDXLUNet._project_context_ahead)
            k, v, ready = kv
            if ready is not None:
                torch.cuda.current_stream().wait_event(ready)
                k.record_stream(torch.cuda.current_stream())
                v.record_stream(torch.cuda.current_stream())
        else:
            k, v = a.to_k(context), a.to_v(context)                 # K/V: BOS path
        x, f3 = a.cross_attend_out(fq, k, v, x, next_ln=ln3)        # x + attn2(norm2(x), ctx)
        (ff,) = f3 if f3 is not None else _ln_feed(self.norm3, x, [self.ff.net[0].proj])
        if next_ln is None:
            return self.ff.forward_fused(ff, x)                     # x + ff(norm3(x))
        return self.ff.forward_fused(ff, x, next_ln=next_ln)        # ... and the next block's norm1 feeds


class Transformer2DModel(nn.Module):
    def __init__(self, dim, depth, cross_dim, head_dim, groups):
        super().__init__()
        self.norm = nn.GroupNorm(groups, dim, eps=1e-6)
        self.proj_in = nn.Linear(dim, dim)
        self.transformer_blocks = nn.ModuleList(
            [BasicTransformerBlock(dim, cross_dim, head_dim) for _ in range(depth)])
        self.proj_out = nn.Linear(dim, dim)

    fused = False

    def forward(self, x, context):
        B, C, H, W = x.shape
        res = x
        if self.fused and _fusable_f16(x) and x.is_contiguous(memory_format=torch.channels_last):
            feed, q = _gn_feed(self.norm, x, self.proj_in, silu=False)
            feed = feed.permute(0, 2, 3, 1).reshape(B, H * W, C)    # NHWC memory == [B, HW, C]
            blocks = list(self.transformer_blocks)
            owner = self.__dict__.get("_ln_owner")
            chain = (LN_CHAIN and owner is not None and not DEFUSE and blocks
                     and all(getattr(b, "fused", False) for b in blocks))
            if chain:
                # every LayerNorm rides in the launch of the GEMM that produces its input (round 5):
                # proj_in -> block 0's norm1, attn1.to_out.0 -> norm2, attn2.to_out.0 -> norm3,
                # ff.net.2 -> the next block's norm1
                first = (blocks[0].norm1, blocks[0].ln1_consumers(), owner)
                if q and _accel(self.proj_in):
                    h, feeds = _gemm_res_ln(self.proj_in, feed, None, first)
                else:
                    h = _run(self.proj_in, (feed, q))
                    feeds = None
                for i, blk in enumerate(blocks):
                    nxt = blocks[i + 1] if i + 1 < len(blocks) else None
                    nl = None if nxt is None else (nxt.norm1, nxt.ln1_consumers(), owner)
                    out = blk.forward_fused(h, context, feeds1=feeds, next_ln=nl, owner=owner)
                    h, feeds = out if nl is not None else (out, None)
            else:
                h = _run(self.proj_in, (feed, q))
                for blk in blocks:
                    h = blk(h, context)
            res_rows = res.permute(0, 2, 3, 1).reshape(B, H * W, C)
            h = _linear_res(self.proj_out, h, res_rows)             # proj_out(h) + res
            return h.reshape(B, H, W, C).permute(0, 3, 1, 2)
        h = self.norm(x).permute(0, 2, 3, 1).reshape(B, H * W, C)   # free when channels-last
        h = self.proj_in(h)
        for blk in self.transformer_blocks:
            h = blk(h, context)
        h = self.proj_out(h)
        h = h.reshape(B, H, W, C).permute(0, 3, 1, 2)               # channels-last view
        return h + res


class Downsample2D(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.conv = nn.Conv2d(c, c, 3, 2, 1)

    def forward(self, x):
        return self.conv(x)


class Upsample2D(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.conv = nn.Conv2d(c, c, 3, 1, 1)

    fused = False

    def forward(self, x):
        conv = self.conv
        if (self.fused and not DEFUSE and _fusable_f16(x) and _accel(conv)
                and x.is_contiguous(memory_format=torch.channels_last)
                and conv.upsample2x_supported(tuple(x.shape))):
            # quantizing commutes with nearest upsampling: quantize the SMALL tensor, and let the
            # conv's halo loader read pixel (y / 2, x / 2) -- no upsampled tensor, fp16 or int8
            from mixdq_amd.nn.Conv2d import quant_op
            return conv.forward_quantized(quant_op(x, *_qp(conv)), upsample2x=True)
        return self.conv(F.interpolate(x, scale_factor=2.0, mode="nearest"))


class DownBlock(nn.Module):
    def __init__(self, cin, cout, temb, n_layers, depth, cfg, add_down):
        super().__init__()
        self.resnets = nn.ModuleList(
            [ResnetBlock2D(cin if i == 0 else cout, cout, temb, cfg["norm_num_groups"])
             for i in range(n_layers)])
        if depth:
            self.attentions = nn.ModuleList(
                [Transformer2DModel(cout, depth, cfg["cross_attention_dim"], cfg["head_dim"],
                                    cfg["norm_num_groups"]) for _ in range(n_layers)])
        self.has_attn = bool(depth)
        if add_down:
            self.downsamplers = nn.ModuleList([Downsample2D(cout)])
        self.has_down = add_down

    def forward(self, x, temb, context):
        outs = []
        for i, res in enumerate(self.resnets):
            x = res(x, temb)
            if self.has_attn:
                x = self.attentions[i](x, context)
            outs.append(x)
        if self.has_down:
            x = self.downsamplers[0](x)
            outs.append(x)
        return x, outs


class MidBlock(nn.Module):
    def __init__(self, c, temb, depth, cfg):
        super().__init__()
        self.attentions = nn.ModuleList(
            [Transformer2DModel(c, depth, cfg["cross_attention_dim"], cfg["head_dim"],
                                cfg["norm_num_groups"])])
        self.resnets = nn.ModuleList([ResnetBlock2D(c, c, temb, cfg["norm_num_groups"])
                                      for _ in range(2)])

    def forward(self, x, temb, context):
        x = self.resnets[0](x, temb)
        x = self.attentions[0](x, context)
        return self.resnets[1](x, temb)


class UpBlock(nn.Module):
    def __init__(self, prev_c, cout, skip_channels, temb, depth, cfg, add_up):
        super().__init__()
        res = []
        for i, skip in enumerate(skip_channels):
            hidden = prev_c if i == 0 else cout
            res.append(ResnetBlock2D(hidden + skip, cout, temb, cfg["norm_num_groups"],
                                     split=hidden))
        self.resnets = nn.ModuleList(res)
        if depth:
            self.attentions = nn.ModuleList(
                [Transformer2DModel(cout, depth, cfg["cross_attention_dim"], cfg["head_dim"],
                                    cfg["norm_num_groups"]) for _ in skip_channels])
        self.has_attn = bool(depth)
        if add_up:
            self.upsamplers = nn.ModuleList([Upsample2D(cout)])
        self.has_up = add_up

    def forward(self, x, skips, temb, context):
        for i, res in enumerate(self.resnets):
            x = res(x, temb, skips.pop())      # input = cat([x, skip], dim=1)
            if self.has_attn:
                x = self.attentions[i](x, context)
        if self.has_up:
            x = self.upsamplers[0](x)
        return x


class SDXLUNet(nn.Module):
    """forward(sample, timestep, encoder_hidden_states, added_cond_kwargs, return_dict=False)
    -> (noise_pred,), the call shape of quantize_sdxl.py:375-385."""

    def __init__(self, cfg=None):
        super().__init__()
        cfg = dict(SDXL_CONFIG, **(cfg or {}))
        self.cfg = cfg
        boc = cfg["block_out_channels"]
        temb = cfg["time_embed_dim"]
        self.conv_in = nn.Conv2d(cfg["in_channels"], boc[0], 3, 1, 1)
        self.time_embedding = TimestepEmbedding(boc[0], temb)
        self.add_embedding = TimestepEmbedding(cfg["projection_class_embeddings_input_dim"], temb)
        n = cfg["layers_per_block"]
        depths = cfg["transformer_layers_per_block"]
        self.down_blocks = nn.ModuleList()
        skip = [boc[0]]
        cin = boc[0]
        for i, cout in enumerate(boc):
            last = i == len(boc) - 1
            self.down_blocks.append(DownBlock(cin, cout, temb, n, depths[i], cfg, not last))
            skip += [cout] * n + ([] if last else [cout])
            cin = cout
        self.mid_block = MidBlock(boc[-1], temb, cfg["mid_transformer_layers"], cfg)
        self.up_blocks = nn.ModuleList()
        prev = boc[-1]
        for i, cout in enumerate(reversed(boc)):
            last = i == len(boc) - 1
            sk = [skip.pop() for _ in range(n + 1)]
            self.up_blocks.append(UpBlock(prev, cout, sk, temb, list(reversed(depths))[i], cfg,
                                          not last))
            prev = cout
        self.conv_norm_out = nn.GroupNorm(cfg["norm_num_groups"], boc[0], eps=1e-5)
        self.conv_act = nn.SiLU()          # (diffusers' name)
        self.conv_out = nn.Conv2d(boc[0], cfg["out_channels"], 3, 1, 1)
        self.register_load_state_dict_post_hook(_refresh_after_load)

    fused = False

    def _grouped(self, name, key, members, w4):
        """Cached device table of a grouped launch (rebuilt when any member's storage changed)."""
        from mixdq_amd._C import GemmGroupTable
        tables = self.__dict__.setdefault("_group_tables", {})
        want = tuple(m[0].data_ptr() for m in members) + tuple(m[4].data_ptr() for m in members)
        t = tables.get((name, key))
        if t is None or t.key != want or t.w4 != bool(w4):
            t = tables[(name, key)] = GemmGroupTable(members, w4=w4)
        return t

    def _project_context_ahead(self, context):
        """Cross-attention keys / values depend only on the text embeddings, not on the latent: in
        the fused graph all 2 x 70 to_k / to_v projections are issued up front -- as ONE grouped
        launch per activation quantizer (gridDim.y = block; 367 MB of weights streamed at HBM rate
        instead of 70 latency-bound launches), each writing its block's persistent [B, T, 2C] buffer
        whose BOS row is a constant.  The INT8 copy of the context (tokens 1.., BOS carve-out) is
        shared by every layer whose activation quantizer is identical.  (Round 1 ran these on a
        side stream; measured, a second long-lived branch in the hipGraph costs more than the
        kernels it hides: 14.96 -> 14.36 ms with both side branches removed.)"""
        blocks = [m for m in self.modules() if isinstance(m, BasicTransformerBlock)]
        if not blocks or not context.is_cuda:
            return
        f16_tail = context.dtype == torch.float16 and context.shape[1] > 1
        B, T = context.shape[0], context.shape[1]

        def bos_w8a8(layer):
            return bool(getattr(layer, "valid_for_acceleration", False)
                        and getattr(layer, "bos", False) and f16_tail)

        # quantizer grouping of every BOS W8A8 K/V layer, taken once (memoised; no host
        # comparison ever runs inside the forward proper, so the loop below is capture-safe)
        kv_layers = [l for blk in blocks for l in (blk.attn2.to_k, blk.attn2.to_v) if bos_w8a8(l)]
        gid = dict(zip(map(id, kv_layers), _quantizer_groups(_memo(self), "ctx", kv_layers)))
        shared = {}                       # quantizer group -> int8 context (tokens 1..)

        def ctx_int8(layer):
            g = gid[id(layer)]
            if g not in shared:
                from mixdq_amd.nn.Linear import quant_op
                shared[g] = quant_op(context[:, 1:, :], layer.act_scales_inv,
                                     layer.act_zero_points)
            return shared[g]

        grouped = {}                      # (quantizer group, K, w4) -> [(block, pack)]
        for blk in blocks:
            lk, lv = blk.attn2.to_k, blk.attn2.to_v
            pack = None
            if bos_w8a8(lk) and bos_w8a8(lv) and gid[id(lk)] == gid[id(lv)]:
                pack = self._kv_pack(blk, context)
            if pack is not None:        # to_k | to_v as ONE GEMM against [2C, K] (cf. _qkv_fused)
                # persistent [B, T, 2C] buffer PER SHAPE (a captured graph keeps raw pointers into
                # it: a buffer is never dropped while its pack lives), BOS row written once -- it is
                # a constant -- and rewritten in place if the BOS tensors changed
                bos_key = (lk.bos_pre_computed.data_ptr(), lk.bos_pre_computed._version,
                           lv.bos_pre_computed.data_ptr(), lv.bos_pre_computed._version)
                slot = pack.setdefault("outs", {}).get((B, T, context.device))
                if slot is None:
                    slot = pack["outs"][(B, T, context.device)] = [None, torch.empty(
                        (B, T, 2 * pack["C"]), dtype=torch.float16, device=context.device)]
                if slot[0] != bos_key:
                    slot[1][:, :1, :] = torch.cat([lk.bos_pre_computed, lv.bos_pre_computed], dim=-1)
                    slot[0] = bos_key
                o = slot[1]
                blk._kv = (o[..., :pack["C"]], o[..., pack["C"]:], None)
                grouped.setdefault((gid[id(lk)], lk.in_features, pack["w4"]), []).append(
                    (lk, pack, o))
                continue
            outs = []
            for layer in (lk, lv):
                if not bos_w8a8(layer):
                    outs.append(layer(context))
                    continue
                # persistent K / V buffer per layer and shape (never dropped: see above)
                bos_key = (layer.bos_pre_computed.data_ptr(), layer.bos_pre_computed._version)
                bufs = layer.__dict__.setdefault("_kv_buf", {})
                slot = bufs.get((B, T, context.device))
                if slot is None:
                    slot = bufs[(B, T, context.device)] = [None, torch.empty(
                        (B, T, layer.out_features), dtype=torch.float16, device=context.device)]
                if slot[0] != bos_key:
                    slot[1][:, :1, :] = layer.bos_pre_computed
                    slot[0] = bos_key
                outs.append(layer.forward_bos_quantized(ctx_int8(layer), B, T, out=slot[1]))
            blk._kv = (outs[0], outs[1], None)
        from mixdq_amd import _C
        from mixdq_amd.op.qlinear import qlinear
        for (g, K, w4), members in grouped.items():
            lk0 = members[0][0]
            x_int = ctx_int8(lk0)
            if len(members) == 1:
                pack, o = members[0][1], members[0][2]
                qlinear(x_int, pack["w"], pack["wscale"], lk0.act_scales, lk0.act_zero_points,
                        pack["wsum"], pack["scale"], pack["bias0"], None, _out=o,
                        _row_map=(T - 1, T, 1), _w4=w4)
                continue
            table = self._grouped("kv", (g, K, w4, B, T), [
                (pk["w"], pk["bias0"], pk["scale"], None, o) for _, pk, o in members], w4)
            _C.qlinear_grouped(x_int, table, _row_map=(T - 1, T, 1))

    @staticmethod
    def _kv_pack(blk, context=None):
        """Cross-attention to_k / to_v read the same tokens; when both are W8A8 BOS layers with the
        same activation quantizer and no bias, one GEMM against the row-concatenated weights
        replaces two (`_pack_rows`: every output element is computed exactly as before)."""
        lk, lv = blk.attn2.to_k, blk.attn2.to_v
        layers = [lk, lv]
        if not (all(getattr(m, "valid_for_acceleration", False) and getattr(m, "bos", False)
                    and m.bias is None for m in layers)
                and lk.out_features == lv.out_features):
            return None
        if context is not None and not (context.dtype == torch.float16 and context.shape[1] > 1):
            return None
        if len(set(_quantizer_groups(_memo(blk), "kv", layers))) != 1:
            return None
        if not _uniform_storage(layers):
            return None                         # mixed int8 / packed storage: see prepare_fused_
        pack = blk.__dict__.get("_kvpack")
        if not _pack_valid(pack, layers):
            pack = blk.__dict__["_kvpack"] = _pack_rows(layers)
        return pack

    def _project_temb_ahead(self, emb):
        """Every ResnetBlock2D adds time_emb_proj(silu(emb)): 22 M = batch GEMMs (plus their SiLU
        and quantize launches) that depend only on the time embedding.  In the fused graph they
        are issued up front: SiLU once, one INT8 copy per distinct activation quantizer (the layers
        are all calibrated on this same tensor) and ONE grouped launch for all layers that share
        it (gridDim.y = layer; N differs per layer), into persistent [B, Cout] buffers."""
        resnets = [m for m in self.modules() if isinstance(m, ResnetBlock2D)]
        if not resnets or not emb.is_cuda:
            return
        s = F.silu(emb)
        B = s.shape[0]
        acc = [r.time_emb_proj for r in resnets if _accel(r.time_emb_proj)]
        ok = _fusable_f16(s) and s.dim() == 2
        gid = dict(zip(map(id, acc), _quantizer_groups(_memo(self), "temb", acc)))
        shared, grouped = {}, {}
        for res in resnets:
            layer = res.time_emb_proj
            if not (_accel(layer) and ok):
                res._t = (layer(s), None)
                continue
            # persistent output per batch size: a graph captured at another batch keeps raw
            # pointers to ITS buffer, so none is ever dropped or reused for a different shape
            bufs = layer.__dict__.setdefault("_t_out", {})
            buf = bufs.get((B, s.device))
            if buf is None:
                buf = bufs[(B, s.device)] = torch.empty((B, layer.out_features),
                                                        dtype=torch.float16, device=s.device)
            res._t = (buf, None)
            grouped.setdefault((gid[id(layer)], layer.in_features, bool(layer.w_packed4)),
                               []).append((layer, buf))
        from mixdq_amd import _C
        for (g, K, w4), members in grouped.items():
            from mixdq_amd.nn.Linear import quant_op
            if g not in shared:
                shared[g] = quant_op(s, *_qp(members[0][0]))
            x_int = shared[g]
            if len(members) == 1:
                members[0][0]._gemm(x_int, out=members[0][1])
                continue
            table = self._grouped("temb", (g, K, w4, B), [
                (m.weight_int4 if w4 else m.weight_int, m.bias0, m.scale, m.bias, buf)
                for m, buf in members], w4)
            _C.qlinear_grouped(x_int, table)

    def refresh_derived_(self):
        """After buffers were written IN PLACE (load_state_dict, shard.broadcast_module_state):
        re-derive, at their existing addresses, the cached tensors that are computed from buffers
        rather than views of them -- the conv border tables and the BOS row of the persistent
        K / V buffers -- so that captured graphs see the new values too.  (Packed GEMM operands
        are views and need nothing.)"""
        with torch.no_grad():
            for m in self.modules():
                if m is not self and hasattr(m, "refresh_derived_"):
                    m.refresh_derived_()
                bufs = m.__dict__.get("_kv_buf")
                if bufs and torch.is_tensor(getattr(m, "bos_pre_computed", None)):
                    for slot in bufs.values():
                        slot[1][:, :1, :] = m.bos_pre_computed.to(slot[1].device)
                        slot[0] = None          # re-keyed on the next forward
                pack = m.__dict__.get("_kvpack")
                if pack is not None and pack.get("outs"):
                    lk, lv = (r() for r in pack["layers"])
                    if lk is None or lv is None:
                        continue
                    for slot in pack["outs"].values():
                        slot[1][:, :1, :] = torch.cat(
                            [lk.bos_pre_computed, lv.bos_pre_computed], dim=-1).to(slot[1].device)
                        slot[0] = None
        return self

    def prepare_fused_(self):
        """Explicit post-conversion step of the fused graph (called by `set_fused(True)`): whatever
        changes a module's BUFFERS happens here, once, never inside a forward -- mixed int8 / packed
        4-bit q|k|v and k|v groups are widened to one storage kind (`unify_packed_storage_`) and
        their row-concatenated GEMM operands are built (`_pack_rows`: the layers' buffers become
        views of the pack).  After it, state-dict keys and shapes are final: a checkpoint saved now
        loads strictly later, and every rank holds the same set of buffers before
        `shard.broadcast_module_state`."""
        for blk in self.modules():
            if not isinstance(blk, BasicTransformerBlock):
                continue
            a = blk.attn1
            qkv = [a.to_q, a.to_k, a.to_v]
            if (all(_accel(m) and m.bias is None for m in qkv)
                    and len(set(_quantizer_groups(_memo(blk), "qkv", qkv))) == 1
                    and unify_packed_storage_(qkv)):
                blk._qkv_fused()
            kv = [blk.attn2.to_k, blk.attn2.to_v]
            if (all(getattr(m, "valid_for_acceleration", False) and getattr(m, "bos", False)
                    and m.bias is None for m in kv)
                    and kv[0].out_features == kv[1].out_features
                    and len(set(_quantizer_groups(_memo(blk), "kv", kv))) == 1
                    and unify_packed_storage_(kv)):
                self._kv_pack(blk)
        for m in self.modules():            # the network that keeps the GEMM + LayerNorm exchange buffer
            if isinstance(m, Transformer2DModel):
                m.__dict__["_ln_owner"] = self
        # the GELU table of the GEMM+GEGLU launches is built here, eagerly: a first use inside a stream
        # capture could only RECORD its init kernel into that graph (csrc/igemm_kernel.h ensure_gelu_table)
        dev = next((b.device for b in self.buffers() if b.is_cuda), None)
        if dev is not None:
            from mixdq_amd import _C
            _C.gelu_table(dev)
            _C.silu_table(dev)              # ... and the SiLU table of the large GroupNorm apply launches (fused_norm.hip)
            if WEIGHT_ARENA:
                # every static tensor of the converted network in ONE allocation, in module order (arena.py)
                from mixdq_amd.arena import pack_static_
                self.__dict__["_arena"] = pack_static_(self, dev)
                self.__dict__.pop("_pf_plans", None)
        return self

    def set_fused(self, enabled: bool = True):
        """Switch the producer fusions on or off for the whole graph (see the top of this file)."""
        self.__dict__.pop("_pf_plans", None)
        for m in self.modules():
            if hasattr(type(m), "fused"):
                m.fused = bool(enabled)
            if isinstance(m, FeedForward):
                m.set_interleaved(bool(enabled))
        if enabled:
            self.prepare_fused_()
        return self

    def forward(self, sample, timestep, encoder_hidden_states, added_cond_kwargs=None,
                return_dict=False):
        cfg = self.cfg
        B = sample.shape[0]
        dtype = sample.dtype
        t = timestep
        if not torch.is_tensor(t):
            t = torch.tensor([t], dtype=torch.float32, device=sample.device)
        t = t.reshape(-1).expand(B)
        emb = self.time_embedding(sinusoidal_embedding(t, cfg["block_out_channels"][0]).to(dtype))
        time_ids = added_cond_kwargs["time_ids"]
        text_embeds = added_cond_kwargs["text_embeds"]
        tid = sinusoidal_embedding(time_ids.flatten(), cfg["addition_time_embed_dim"])
        add = torch.cat([text_embeds, tid.reshape(B, -1).to(dtype)], dim=-1)
        emb = emb + self.add_embedding(add)

        use_pf = bool(self.fused and PREFETCH and sample.is_cuda and _fusable_f16(sample) and not DEFUSE)
        if not use_pf:
            return self._forward_body(sample, emb, encoder_hidden_states)
        # Trace this forward's weight order and replay the plan of the last forward ON THIS DEVICE: the state
        # is this call's own PrefetchContext (thread-local while it runs), the plans are kept per device in
        # the instance and hold weak references -- a plan made for another device, or for weights that have
        # been replaced since (`.to()`, re-quantization, load_state_dict: cleared in _apply / below, and
        # dropped here when a reference is dead or points elsewhere), is never handed to a launch.
        from mixdq_amd import _C
        plans = self.__dict__.setdefault("_pf_plans", {})
        plan = plans.get(sample.device)
        if plan is not None and not _plan_alive(plan, sample.device):
            plans.pop(sample.device, None)
            plan = None
        ctx = _C.PrefetchContext(sample.device, plan["lists"] if plan else None)
        try:
            with ctx:
                return self._forward_body(sample, emb, encoder_hidden_states)
        finally:
            if ctx._outer is None:                               # (an inner, inactive context traced nothing)
                sig = _trace_signature(ctx.trace)
                if plan is None or plan["sig"] != sig:           # first forward here, or the layers changed
                    new = _build_prefetch_plan(ctx.trace)
                    if new is None:
                        plans.pop(sample.device, None)
                    else:
                        plans[sample.device] = new

    def _apply(self, fn, recurse=True):          # .to() / .cuda() / .half(): the weights move, the plans go
        self.__dict__.pop("_pf_plans", None)
        return super()._apply(fn, recurse)

    def load_state_dict(self, *args, **kwargs):
        self.__dict__.pop("_pf_plans", None)
        return super().load_state_dict(*args, **kwargs)

    def _forward_body(self, sample, emb, encoder_hidden_states):
        if self.fused and _fusable_f16(sample) and not DEFUSE:
            self._project_temb_ahead(emb)
            self._project_context_ahead(encoder_hidden_states)
        x = sample.contiguous(memory_format=torch.channels_last)
        x = self.conv_in(x)
        skips = [x]
        for blk in self.down_blocks:
            x, outs = blk(x, emb, encoder_hidden_states)
            skips += outs
        x = self.mid_block(x, emb, encoder_hidden_states)
        for blk in self.up_blocks:
            x = blk(x, skips, emb, encoder_hidden_states)
        if self.fused and _fusable_f16(x):
            feed, q = _gn_feed(self.conv_norm_out, x, self.conv_out, silu=True)
            x = self.conv_out.forward_quantized(feed) if q else self.conv_out(feed)
        else:
            x = self.conv_out(self.conv_act(self.conv_norm_out(x)))
        return (x,)


# ---------------------------------------------------------------------------------------------
# synthetic weights, layer inventory
# ---------------------------------------------------------------------------------------------
def quantizable_layers(unet: nn.Module) -> "OrderedDict[str, nn.Module]":
    """name -> nn.Linear / nn.Conv2d in named_modules() order: the keys of the bit-width yamls."""
    return OrderedDict((n, m) for n, m in unet.named_modules()
                       if isinstance(m, (nn.Linear, nn.Conv2d)))


@torch.no_grad()
def init_synthetic_weights(unet: nn.Module, seed: int = 42, std: float = 0.02):
    """randn * 0.02, generator seeded with seed + layer index."""
    for idx, (name, mod) in enumerate(quantizable_layers(unet).items()):
        g = torch.Generator(device="cpu").manual_seed(seed + idx)
        mod.weight.copy_(torch.randn(mod.weight.shape, generator=g) * std)
        if mod.bias is not None:
            mod.bias.copy_(torch.randn(mod.bias.shape, generator=g) * std)
    return unet


def build_unet(device=None, dtype=torch.float16, seed: int = 42, cfg=None) -> SDXLUNet:
    unet = SDXLUNet(cfg)
    init_synthetic_weights(unet, seed)
    unet = unet.to(dtype=dtype)
    if device is not None:
        unet = unet.to(device)
    unet = unet.to(memory_format=torch.channels_last)
    return unet.eval()
