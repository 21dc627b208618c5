"""QuantizedConv2d for MI355X: counterpart of mixdq_extension/nn/Conv2d.py:16-347.

Same constructor, `from_float(float_mod, split=0, ckpt=None)`, buffer names (incl. the `_0`
buffers of a split up-block conv_shortcut) and forward semantics.

Differences, all result-preserving:
  * for padded convs the zero-point border table (include/mixdq_hip.h mixdq_conv_border_table)
    depends only on the weights, so it is built once and cached next to
    weight_sum_by_input_channels instead of recomputing an [N,P,Q,K] f32 tensor per call
    (qconv2d.cc:131-136);
  * the channel slices x[:, :split], x[:, split:] are quantized with their real strides (the
    reference reads them linearly: right only at batch 1 / NCHW -- SURVEY.md section 0).
"""
from __future__ import annotations

import logging

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.ao.quantization import QConfig

from mixdq_amd import _C
from mixdq_amd.nn.glue import tagged_operand
from mixdq_amd.nn.utils import create_qparams_from_dtype, pack_w4, unpack_w4
from mixdq_amd.op.quant import quantize_per_tensor_vectorized

__all__ = ["QuantizedConv2d"]

quant_op = quantize_per_tensor_vectorized

_INT8 = (torch.qint8, torch.quint8)


def _w8a8_ok(w, a, w4_kernel=False) -> bool:
    w_ok = w is not None and (w.dtype in _INT8 or (w4_kernel and w.dtype == torch.quint4x2))
    return bool(
        w_ok and a is not None and a.dtype in _INT8
        and w.qscheme == torch.per_channel_affine and a.qscheme == torch.per_tensor_affine
        and torch.all(w.zero_points == 0.0).item())


def _refresh_after_load(mod, _incompatible_keys):
    mod.refresh_derived_()


class QuantizedConv2d(nn.Module):
    w4_kernel = False   # W4A8 kernel path, see QuantizedLinear.w4_kernel

    def __init__(self, in_channels: int, out_channels: int, kernel_size, stride, padding,
                 dilation, groups=1, bias=True, device=None, w_qparams=None, w_qparams_0=None,
                 a_qparams=None, a_qparams_0=None, module_name=None, split=0,
                 w4_kernel=False) -> None:
        super().__init__()
        self.w_packed4 = bool(w4_kernel and w_qparams is not None
                              and w_qparams.dtype == torch.quint4x2)
        self.module_name = module_name
        self.split = split   # > 0: up-block conv_shortcut, input = cat(hidden[:split], skip)
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.device = device
        self.kernel_size = kernel_size
        self.stride = stride
        self.padding = padding
        self.dilation = dilation
        self.groups = groups
        square = (len(set(stride)) == 1 and len(set(padding)) == 1 and len(set(dilation)) == 1
                  and dilation[0] == 1 and groups == 1)
        self.valid_for_acceleration = (
            _w8a8_ok(w_qparams, a_qparams, w4_kernel)
            and (split == 0 or _w8a8_ok(w_qparams_0, a_qparams_0, w4_kernel))
            and square)
        if self.valid_for_acceleration and self.w_packed4:
            cins = [in_channels] if split == 0 else [split, in_channels - split]
            if any(c % 32 != 0 for c in cins):       # packed pieces span 32 input channels
                self.valid_for_acceleration = False
        if self.valid_for_acceleration and (in_channels % 4 != 0 or out_channels % 4 != 0):
            logging.warning(
                f"Conv2d layer with in_channels = {in_channels} and out_channels = "
                f"{out_channels} cannot use quantized kernel due to misalignment. "
                "Falling back to FP kernels")
            self.valid_for_acceleration = False
        if self.valid_for_acceleration:
            self._register_qparams("", w_qparams, a_qparams, device)
            if split != 0:
                self._register_qparams("_0", w_qparams_0, a_qparams_0, device)
            self.register_load_state_dict_post_hook(_refresh_after_load)

    def _register_qparams(self, sfx, w, a, device):
        self.register_buffer("weight_scales" + sfx, w.scales.to(device).float())
        self.register_buffer("weight_zero_points" + sfx, w.zero_points.to(device).float())
        self.register_buffer("act_scales" + sfx, a.scales.to(device).float())
        self.register_buffer("act_zero_points" + sfx, a.zero_points.to(device).float())
        self.register_buffer("act_scales_inv" + sfx, 1 / getattr(self, "act_scales" + sfx))

    def _register_weight(self, sfx, weight, w_qparams, pad):
        scales = getattr(self, "weight_scales" + sfx)
        azp = getattr(self, "act_zero_points" + sfx)
        if self.w_packed4:   # the Path A integers, packed along C in [K, R, S, C] order
            lim = 2 ** (self.w_bit - 1)          # 2-bit layers: [-2, 1] in 4-bit storage
            weight_int = torch.clamp(torch.round(weight.float() / scales[:, None, None, None]),
                                     -lim, lim - 1).to(torch.int8)
            packed = pack_w4(weight_int.permute(0, 2, 3, 1).contiguous())      # [K, R, S, C/2]
            self.register_buffer("weight_int4" + sfx, packed.permute(0, 3, 1, 2))
        else:
            weight_int = torch.quantize_per_channel(
                weight.float(), scales, getattr(self, "weight_zero_points" + sfx),
                axis=w_qparams.axis, dtype=w_qparams.dtype).int_repr()
            # KRSC in memory (the layout the kernel reads): stored once, never re-laid-out per call
            weight_int = weight_int.contiguous(memory_format=torch.channels_last)
            self.register_buffer("weight_int" + sfx, weight_int)
        if pad == 0:
            # per-channel zero-point term (nn/Conv2d.py:166-171)
            self.register_buffer("bias0" + sfx, weight_int.float().sum(dim=[1, 2, 3]) * azp)
            setattr(self, "weight_sum_by_input_channels" + sfx, None)
        else:
            # per-tap sums; the border-dependent term is applied in the kernel (:172-177)
            self.register_buffer("weight_sum_by_input_channels" + sfx,
                                 weight_int.float().sum(dim=1, keepdim=True))
            setattr(self, "bias0" + sfx, None)
        self.register_buffer("scale" + sfx, scales * getattr(self, "act_scales" + sfx))

    @classmethod
    def from_float(cls, float_mod, split=0, ckpt=None):
        assert hasattr(float_mod, "qconfig") and isinstance(float_mod.qconfig, QConfig)
        w_dtype = float_mod.qconfig.weight().dtype
        a_dtype = float_mod.qconfig.activation().dtype
        device = float_mod.weight.device
        common = dict(device=device, num_kernels=float_mod.weight.shape[0], ckpt=ckpt,
                      module_name=float_mod.module_name, split=split)
        w_qparams, w_qparams_0 = create_qparams_from_dtype(
            dtype=w_dtype, is_channel_wise=True, quant_type="weight",
            bit_width=float_mod.w_bit, **common)
        a_qparams = a_qparams_0 = None
        if hasattr(float_mod, "a_bit"):
            a_qparams, a_qparams_0 = create_qparams_from_dtype(
                dtype=a_dtype, is_channel_wise=False, quant_type="act",
                bit_width=float_mod.a_bit, **common)
        new_mod = cls(float_mod.in_channels, float_mod.out_channels, float_mod.kernel_size,
                      float_mod.stride, float_mod.padding, float_mod.dilation, float_mod.groups,
                      float_mod.bias is not None, device=device, w_qparams=w_qparams,
                      w_qparams_0=w_qparams_0, a_qparams=a_qparams, a_qparams_0=a_qparams_0,
                      module_name=float_mod.module_name, split=split,
                      w4_kernel=getattr(float_mod, "w4_kernel", cls.w4_kernel))
        new_mod.w_bit = int(getattr(float_mod, "w_bit", 8))
        weight = float_mod.weight.detach()
        pad = float_mod.padding[0]
        if new_mod.valid_for_acceleration:
            if split == 0:
                new_mod._register_weight("", weight, w_qparams, pad)
            else:
                new_mod._register_weight("", weight[:, :split, ...], w_qparams, pad)
                new_mod._register_weight("_0", weight[:, split:, ...], w_qparams_0, pad)
        else:
            new_mod.register_buffer("weight", weight)
        if float_mod.bias is not None:
            new_mod.register_buffer("bias", float_mod.bias.detach())
        else:
            new_mod.bias = None
        return new_mod

    def _get_name(self):
        if not self.valid_for_acceleration:
            return "QuantizedConv2dFPFallback"
        return "QuantizedConv2dW4A8" if self.w_packed4 else "QuantizedConv2dW8A8"

    def _weight_values(self, sfx=""):
        if not self.w_packed4:
            return getattr(self, "weight_int" + sfx)
        p = getattr(self, "weight_int4" + sfx).permute(0, 2, 3, 1).contiguous()   # [K,R,S,C/2]
        return unpack_w4(p).permute(0, 3, 1, 2)

    def forward_fallback(self, x: torch.Tensor):
        def deq(sfx):
            w = self._weight_values(sfx).float()
            return (w * getattr(self, "weight_scales" + sfx)[:, None, None, None]).to(x.dtype)

        bias = self.bias.to(x.dtype) if self.bias is not None else None
        args = (self.stride, self.padding, self.dilation, self.groups)
        if self.split == 0:
            return F.conv2d(x, deq(""), bias, *args)
        return (F.conv2d(x[:, :self.split], deq(""), bias, *args)
                + F.conv2d(x[:, self.split:], deq("_0"), None, *args))

    def _border_table(self, sfx):
        """Cached tap-rectangle sums for padded convs.  Derived from the weights only, so it is
        built once -- and rebuilt IN PLACE (same address: captured graphs keep reading it) when
        weight_sum_by_input_channels is replaced or updated in place (load_state_dict, broadcast)."""
        if self.padding[0] == 0:
            return None
        wsum = getattr(self, "weight_sum_by_input_channels" + sfx)
        cache = self.__dict__.setdefault("_tables", {})
        e = cache.get(sfx)
        if e is not None and e[0] is wsum and e[1] == wsum._version:
            return e[2]
        out = None
        if e is not None and e[2].device == wsum.device and e[2].shape[1] == wsum.shape[0]:
            out = e[2]
        table = _C.conv_border_table(wsum, out=out)
        cache[sfx] = (wsum, wsum._version, table)
        return table

    def refresh_derived_(self):
        """Re-derive cached state from the buffers after they were written in place."""
        if self.valid_for_acceleration and self.padding[0] > 0 and self.scale.is_cuda:
            for sfx in ("", "_0") if self.split else ("",):
                self._border_table(sfx)

    def forward_quantized(self, x_int, residual=None, residual_per_image=False, upsample2x=False):
        """The conv half of forward() for an input already quantized with this layer's activation
        qparams (not for split shortcuts).  `residual`: fp16 [N,K,P,Q] channels-last, or [N,K] with
        residual_per_image (the time-embedding add), added after the epilogue's FP16 rounding.
        upsample2x: the conv runs on the nearest 2x upsampling of x_int (never materialised)."""
        assert self.valid_for_acceleration and self.split == 0
        return self._conv(x_int, "", self.bias, residual, residual_per_image, upsample2x)

    def upsample2x_supported(self, x_shape) -> bool:
        return bool(self.valid_for_acceleration and self.split == 0 and not self.w_packed4
                    and _C.conv_upsample2x_supported(x_shape, (self.out_channels, self.in_channels)
                                                     + tuple(self.kernel_size), self.stride[0],
                                                     self.padding[0]))

    def _conv(self, x_int, sfx, bias, residual=None, residual_per_image=False, upsample2x=False):
        return _C.qconv2d_w8_a8_ohalf(
            x_int, getattr(self, ("weight_int4" if self.w_packed4 else "weight_int") + sfx),
            getattr(self, "weight_scales" + sfx),
            getattr(self, "act_scales" + sfx), getattr(self, "act_zero_points" + sfx),
            getattr(self, "scale" + sfx), getattr(self, "weight_sum_by_input_channels" + sfx),
            getattr(self, "bias0" + sfx), bias, self.stride[0], self.padding[0], 1,
            _table=self._border_table(sfx), _residual=residual,
            _residual_per_image=residual_per_image, _w4=self.w_packed4, _upsample2x=upsample2x)

    # FP fallback layers: True = this repo's FP16 MFMA kernel (mixdq_conv2d_f16), False = F.conv2d
    # as in the reference (MIOpen here, whose kernel search costs ~30 s of start-up).
    fp16_kernel = True

    def forward_fp(self, x, residual=None, residual_per_image=False):
        """The reference's FP fallback, F.conv2d on the FP16 weight (nn/Conv2d.py:306-309)."""
        square = (len(set(self.stride)) == 1 and len(set(self.padding)) == 1
                  and all(d == 1 for d in self.dilation) and self.groups == 1)
        if (self.fp16_kernel and square and x.is_cuda and x.dtype == torch.float16
                and self.weight.dtype == torch.float16 and self.out_channels % 4 == 0):
            w = self.weight
            if not w.is_contiguous(memory_format=torch.channels_last):
                # KRSC in memory, once (the kernel's layout; a captured graph must not re-lay it out)
                w = self.weight = w.contiguous(memory_format=torch.channels_last)
            if self.in_channels % 8:
                # conv_in (4 input channels): zero-pad the channels to a 16-byte multiple so the
                # layer runs on the MFMA tiles (the padded weight is cached; zeros add nothing)
                cp = -self.in_channels % 8
                wp = self.__dict__.get("_w_padded")
                if wp is None or wp[0] is not w or wp[1] != w._version:
                    wpad = torch.zeros((w.shape[0], w.shape[1] + cp, w.shape[2], w.shape[3]),
                                       dtype=w.dtype, device=w.device
                                       ).contiguous(memory_format=torch.channels_last)
                    wpad[:, :w.shape[1]] = w
                    wp = self.__dict__["_w_padded"] = (w, w._version, wpad)
                xp = torch.zeros((x.shape[0], x.shape[1] + cp, x.shape[2], x.shape[3]),
                                 dtype=x.dtype, device=x.device
                                 ).contiguous(memory_format=torch.channels_last)
                xp[:, :x.shape[1]] = x
                x, w = xp, wp[2]
            if residual is not None and not residual_per_image and not residual.is_contiguous(
                    memory_format=torch.channels_last):
                return _C.conv2d_f16(x, w, self.bias, self.stride[0], self.padding[0]) + residual
            return _C.conv2d_f16(x, w, self.bias, self.stride[0], self.padding[0],
                                 _residual=residual, _residual_per_image=residual_per_image)
        y = F.conv2d(x, self.weight, self.bias, self.stride, self.padding, self.dilation,
                     self.groups)
        if residual is None:
            return y
        return y + (residual[:, :, None, None] if residual_per_image else residual)

    def forward_parts(self, x_a: torch.Tensor, x_b: torch.Tensor) -> torch.Tensor:
        """forward(cat([x_a, x_b], dim=1)) of a split layer (x_a has `split` channels) without the
        concatenation: each half is quantized with its own activation quantizer where it lies."""
        assert self.valid_for_acceleration and self.split == x_a.shape[1]
        first = self._quant_conv(x_a, "", self.bias)
        return self._quant_conv(x_b, "_0", None, residual=first)

    def forward_parts_quantized(self, x_int: torch.Tensor, x_int_0: torch.Tensor) -> torch.Tensor:
        """forward_parts for halves already quantized with this layer's two activation quantizers
        (act_scales / act_scales_0), e.g. by the GroupNorm pass that reads the same tensors."""
        assert self.valid_for_acceleration and self.split == x_int.shape[1]
        first = self._conv(x_int, "", self.bias)
        return self._conv(x_int_0, "_0", None, residual=first)

    def _pointwise_f16in(self, x, sfx, bias, residual=None):
        """quant_op(x) -> _conv in ONE launch for a 1x1 / stride-1 / pad-0 layer (the UNet's conv_shortcut):
        over NHWC rows it is a Linear, and mixdq_qlinear_f16in_w8a8 quantizes the FP16 rows in its staging
        path -- a channel slice x[:, a:b] of a channels-last tensor is read in place (row stride = all
        channels).  Returns None where that launch does not take the problem (the caller then issues the
        reference's two launches, nn/Conv2d.py:294-311)."""
        if (self.kernel_size[0] != 1 or self.kernel_size[1] != 1 or self.stride[0] != 1
                or self.padding[0] != 0 or x.dim() != 4):
            return None
        rows = x.permute(0, 2, 3, 1)                      # [N, H, W, C] view of the NHWC memory
        w = getattr(self, ("weight_int4" if self.w_packed4 else "weight_int") + sfx)
        w2 = w.permute(0, 2, 3, 1).reshape(w.shape[0], -1)           # [K, C] (or [K, C / 2] packed): a view
        K = self.out_channels
        if not _C.qlinear_f16in_wanted(rows, K, x.shape[1], w4=self.w_packed4):
            return None
        res = None
        if residual is not None:
            if not residual.is_contiguous(memory_format=torch.channels_last):
                return None
            res = residual.permute(0, 2, 3, 1)
        out = _C.qlinear_f16in(rows, getattr(self, "act_scales_inv" + sfx),
                               getattr(self, "act_zero_points" + sfx), w2, getattr(self, "scale" + sfx),
                               getattr(self, "bias0" + sfx), bias, _residual=res, _w4=self.w_packed4,
                               _trace=w)      # (the module's buffer, not the per-call view of it)
        return out.permute(0, 3, 1, 2)                    # [N, K, H, W], channels-last in memory

    def _quant_conv(self, x, sfx, bias, residual=None):
        """The reference's pair per (half of a) layer -- quantize, conv -- as one launch where possible."""
        y = self._pointwise_f16in(x, sfx, bias, residual)
        if y is not None:
            return y
        x_int = quant_op(x, getattr(self, "act_scales_inv" + sfx), getattr(self, "act_zero_points" + sfx))
        return self._conv(x_int, sfx, bias, residual=residual)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if not self.valid_for_acceleration:
            return self.forward_fp(x)
        if x.dtype != torch.float16:
            return self.forward_fallback(x)
        if self.split == 0:
            x_int = tagged_operand(x, self)       # (swap_glue: written by the producer of x -- nn/glue.py)
            if x_int is not None:
                return self._conv(x_int, "", self.bias)
            return self._quant_conv(x, "", self.bias)
        # bias is applied once, in the first half (nn/Conv2d.py:341-343); the reference's half add of
        # the two fp16 outputs rides in the second launch's epilogue (same arithmetic: each output
        # rounded to fp16, then one fp32 add rounded to fp16)
        first = self._quant_conv(x[:, :self.split], "", self.bias)
        return self._quant_conv(x[:, self.split:], "_0", None, residual=first)
