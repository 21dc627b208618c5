"""QuantizedLinear for MI355X: counterpart of mixdq_extension/nn/Linear.py:17-194.

Same constructor, `from_float(float_mod, split=0, ckpt=None)`, buffer names and forward
semantics.  W8A8 layers run  quantize (HIP) -> INT8 GEMM + epilogue (HIP);  anything else
(4/2-bit weights, missing activation qparams, misaligned sizes) keeps the FP16 weight and runs
F.linear, exactly as the reference does (nn/Linear.py:31,37-43,133-134,155-156).

Differences, all result-preserving:
  * the BOS path (attn2.to_k / to_v) quantizes the strided slice x[:, 1:, :] correctly at any
    batch size (the reference reads it linearly, right only at batch 1 -- SURVEY.md section 0) and
    the GEMM writes tokens 1..T-1 straight into the [B, T, N] result next to the precomputed
    token-0 row, instead of torch.cat (nn/Linear.py:190-193).
"""
from __future__ import annotations

import logging

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.ao.quantization import QConfig

from mixdq_amd.nn.glue import tagged_operand
from mixdq_amd.nn.utils import create_qparams_from_dtype, pack_w4, unpack_w4
from mixdq_amd.op.quant import quantize_per_tensor_vectorized
from mixdq_amd.op.qlinear import qlinear

__all__ = ["QuantizedLinear"]

quant_op = quantize_per_tensor_vectorized

_INT8 = (torch.qint8, torch.quint8)


def _w8a8_ok(w_qparams, a_qparams, w4_kernel=False) -> bool:
    """nn/Linear.py:27-36: per-channel symmetric int8 weights, per-tensor int8 activations.
    With w4_kernel (this build's W4A8 path, off by default) 4-bit weights qualify too."""
    w_ok = w_qparams is not None and (
        w_qparams.dtype in _INT8 or (w4_kernel and w_qparams.dtype == torch.quint4x2))
    return bool(
        w_ok and a_qparams is not None and a_qparams.dtype in _INT8
        and w_qparams.qscheme == torch.per_channel_affine
        and a_qparams.qscheme == torch.per_tensor_affine
        and torch.all(w_qparams.zero_points == 0.0).item())


class QuantizedLinear(nn.Module):
    # W4A8 kernel path.  False = the reference's behaviour: 4-/2-bit
    # layers keep their FP16 weight and run F.linear.  True: weights are stored as packed signed
    # 4-bit integers (the Path A integers: clamp(round(w / delta), -8, 7)) and run on the INT8
    # MFMA kernels with an in-kernel unpack.  Set per float module (`mod.w4_kernel = True`) or via
    # quantize_unet(..., w4_kernel=True).
    w4_kernel = False

    def __init__(self, in_features: int, out_features: int, bias: bool = True, device=None,
                 w_qparams=None, a_qparams=None, module_name=None, w4_kernel=False) -> None:
        super().__init__()
        self.module_name = module_name
        self.in_features = in_features
        self.out_features = out_features
        self.device = device
        self.w_packed4 = bool(w4_kernel and w_qparams is not None
                              and w_qparams.dtype == torch.quint4x2)
        self.valid_for_acceleration = _w8a8_ok(w_qparams, a_qparams, w4_kernel)
        if self.valid_for_acceleration and self.w_packed4 and in_features % 32 != 0:
            self.valid_for_acceleration = False      # packed pieces span 32 input channels
        if self.valid_for_acceleration and (in_features % 4 != 0 or out_features % 4 != 0):
            logging.warning(
                f"Linear layer with in_features = {in_features} and out_features = "
                f"{out_features} cannot use quantized kernel due to misalignment. "
                "Falling back to FP kernels")
            self.valid_for_acceleration = False
        if self.valid_for_acceleration:
            self.register_buffer("weight_scales", w_qparams.scales.to(device).float())
            self.register_buffer("weight_zero_points", w_qparams.zero_points.to(device).float())
            self.register_buffer("act_scales", a_qparams.scales.to(device).float())
            self.register_buffer("act_zero_points", a_qparams.zero_points.to(device).float())
            self.register_buffer("act_scales_inv", 1 / self.act_scales)

    @classmethod
    def from_float(cls, float_mod, split=0, ckpt=None):
        assert hasattr(float_mod, "qconfig") and isinstance(float_mod.qconfig, QConfig)
        w_dtype = float_mod.qconfig.weight().dtype
        a_dtype = float_mod.qconfig.activation().dtype
        device = float_mod.weight.device
        common = dict(device=device, num_kernels=float_mod.weight.shape[0], ckpt=ckpt,
                      module_name=float_mod.module_name, split=split)
        w_qparams, _ = create_qparams_from_dtype(dtype=w_dtype, is_channel_wise=True,
                                                 quant_type="weight", bit_width=float_mod.w_bit,
                                                 **common)
        a_qparams = None
        if hasattr(float_mod, "a_bit"):   # no a_bit => activation stays fp16 => FP fallback
            a_qparams, _ = create_qparams_from_dtype(dtype=a_dtype, is_channel_wise=False,
                                                     quant_type="act", bit_width=float_mod.a_bit,
                                                     **common)
        new_mod = cls(float_mod.in_features, float_mod.out_features, float_mod.bias is not None,
                      device=device, w_qparams=w_qparams, a_qparams=a_qparams,
                      module_name=float_mod.module_name,
                      w4_kernel=getattr(float_mod, "w4_kernel", cls.w4_kernel))
        weight = float_mod.weight.detach()
        name = float_mod.module_name
        if "attn2" in name and ("to_k" in name or "to_v" in name):
            new_mod.bos = float_mod.bos
            new_mod.register_buffer("bos_pre_computed", float_mod.bos_pre_computed)
        if new_mod.valid_for_acceleration and new_mod.w_packed4:
            # the Path A integers (base_quantizer.py:119-127: sym, clamp to the layer's own bit
            # width -- a 2-bit layer loads the 2-bit delta and clamps to [-2, 1]; storage is 4-bit)
            lim = 2 ** (int(getattr(float_mod, "w_bit", 4)) - 1)
            weight_int = torch.clamp(torch.round(weight.float() / new_mod.weight_scales[:, None]),
                                     -lim, lim - 1).to(torch.int8)
            new_mod.register_buffer("weight_int4", pack_w4(weight_int))
        elif new_mod.valid_for_acceleration:
            weight_int = torch.quantize_per_channel(
                weight.float(), new_mod.weight_scales, new_mod.weight_zero_points,
                axis=w_qparams.axis, dtype=w_qparams.dtype).int_repr()
            new_mod.register_buffer("weight_int", weight_int)
        if new_mod.valid_for_acceleration:
            # auxiliary vectors of the epilogue  D = (acc - bias0) * scale + bias
            wsum = weight_int.float().sum(dim=1)
            new_mod.register_buffer("weight_sum_by_input_channels", wsum)
            new_mod.register_buffer("scale", new_mod.weight_scales * new_mod.act_scales)
            new_mod.register_buffer("bias0", wsum * new_mod.act_zero_points)
        else:
            new_mod.register_buffer("weight", weight)
        if float_mod.bias is not None:
            new_mod.register_buffer("bias", float_mod.bias.detach())
        else:
            new_mod.bias = None
        return new_mod

    def _get_name(self):
        if not self.valid_for_acceleration:
            return "QuantizedLinearFPFallback"
        return "QuantizedLinearW4A8" if self.w_packed4 else "QuantizedLinearW8A8"

    def _weight_values(self):
        return unpack_w4(self.weight_int4) if self.w_packed4 else self.weight_int

    def forward_fallback(self, x):
        w = (self._weight_values().float() * self.weight_scales[:, None]).to(x.dtype)
        return F.linear(x, w, self.bias.to(x.dtype) if self.bias is not None else None)

    def _gemm(self, x_int, out=None, row_map=None, residual=None):
        if self.w_packed4:
            return qlinear(x_int, self.weight_int4, self.weight_scales, self.act_scales,
                           self.act_zero_points, self.weight_sum_by_input_channels, self.scale,
                           self.bias0, self.bias, _out=out, _row_map=row_map, _residual=residual,
                           _w4=True)
        return qlinear(x_int, self.weight_int, self.weight_scales, self.act_scales,
                       self.act_zero_points, self.weight_sum_by_input_channels, self.scale,
                       self.bias0, self.bias, _out=out, _row_map=row_map, _residual=residual)

    def forward_quantized(self, x_int: torch.Tensor, residual=None) -> torch.Tensor:
        """The GEMM half of forward() for an input that a fused producer already quantized with
        this layer's (act_scales_inv, act_zero_points).  `residual` (fp16, same shape as the
        output) is added after the epilogue's FP16 rounding, as a following `+` would."""
        assert self.valid_for_acceleration and not getattr(self, "bos", False)
        return self._gemm(x_int, residual=residual)

    # per-output-channel tensors, reordered together by permute_output_rows_
    _ROW_TENSORS = ("weight_int", "weight_int4", "weight", "weight_scales", "weight_zero_points",
                    "weight_sum_by_input_channels", "scale", "bias0", "bias")

    @torch.no_grad()
    def permute_output_rows_(self, perm: torch.Tensor):
        """Reorder the output channels IN PLACE (storage addresses stay valid for captured graphs):
        row i of every per-channel tensor becomes the old row perm[i].  Used to store a GEGLU
        projection in the value/gate-interleaved order of the fused GEMM+GEGLU kernel; the caller
        undoes it with the inverse permutation."""
        for name in self._ROW_TENSORS:
            t = getattr(self, name, None)
            if torch.is_tensor(t) and t.dim() >= 1 and t.size(0) == self.out_features:
                t.copy_(t[perm.to(t.device)])

    def forward_quantized_geglu(self, x_int: torch.Tensor, consumer) -> torch.Tensor:
        """For a GEGLU projection whose rows are stored interleaved (unet.FeedForward): int8
        operand of `consumer` (ff.net.2), = consumer's quantizer applied to fp16(value *
        fp16(gelu(gate))) of this layer's fp16 output, in one launch."""
        from mixdq_amd._C import qlinear_geglu
        assert self.valid_for_acceleration and not getattr(self, "bos", False)
        w = self.weight_int4 if self.w_packed4 else self.weight_int
        return qlinear_geglu(x_int, w, self.scale, self.bias0, self.bias, consumer.act_scales_inv,
                             consumer.act_zero_points, _w4=self.w_packed4)

    def forward_bos_quantized(self, x_int_tail: torch.Tensor, B: int, T: int,
                              out: torch.Tensor = None) -> torch.Tensor:
        """BOS-path output [B, T, N] from the already quantized tokens 1..T-1 (int8 [B, T-1, K]).
        `out`: a caller-owned [B, T, N] buffer whose row 0 already holds bos_pre_computed (the row
        never changes, so a caller that keeps the buffer saves the copy): only rows 1.. are written."""
        assert self.valid_for_acceleration and getattr(self, "bos", False)
        if out is None:
            out = torch.empty((B, T, self.out_features), dtype=torch.float16,
                              device=x_int_tail.device)
            out[:, :1, :] = self.bos_pre_computed
        if T > 1:
            self._gemm(x_int_tail, out=out, row_map=(T - 1, T, 1))
        return out

    # FP fallback layers (no activation quantizer / unsupported weight bits): True = this repo's
    # FP16 MFMA kernel (mixdq_linear_f16), False = F.linear as in the reference (hipBLASLt).
    fp16_kernel = True

    def forward_fp(self, x, residual=None):
        """The reference's FP fallback, F.linear(x, weight, bias) (nn/Linear.py:155-156)."""
        if (self.fp16_kernel and x.is_cuda and x.dtype == torch.float16
                and self.weight.dtype == torch.float16 and self.in_features % 8 == 0
                and self.out_features % 4 == 0):
            from mixdq_amd._C import linear_f16
            if residual is not None and not residual.is_contiguous():
                return linear_f16(x, self.weight, self.bias) + residual
            return linear_f16(x, self.weight, self.bias, _residual=residual)
        y = F.linear(x, self.weight, self.bias)
        return y if residual is None else y + residual

    def _gemm_f16in(self, x, out=None, bos=False, residual=None):
        """quant_op(x) -> _gemm in ONE launch (mixdq_qlinear_f16in_w8a8: the GEMM quantizes its FP16
        operand in its staging path; bit-identical to the pair)."""
        from mixdq_amd._C import qlinear_f16in
        w = self.weight_int4 if self.w_packed4 else self.weight_int
        return qlinear_f16in(x, self.act_scales_inv, self.act_zero_points, w, self.scale, self.bias0,
                             self.bias, _out=out, _bos=bos, _residual=residual, _w4=self.w_packed4)

    def forward(self, x: torch.Tensor, _bos_out: torch.Tensor = None) -> torch.Tensor:
        """`_bos_out` (BOS layers): a caller-owned [B, T, N] FP16 buffer whose row 0 already holds
        bos_pre_computed -- the output is written there (rows 1..) and no row-0 copy is launched; for a caller
        that consumes the result before its next call (nn/glue.py: the swapped cross-attention)."""
        if not self.valid_for_acceleration:
            return self.forward_fp(x)
        if x.dtype != torch.float16:
            return self.forward_fallback(x)
        from mixdq_amd._C import qlinear_f16in_wanted
        N, K = self.out_features, self.in_features
        if not getattr(self, "bos", False):
            # (swap_glue: the producer of x already wrote quantize(x) for THIS layer's quantizer -- nn/glue.py)
            x_int = tagged_operand(x, self)
            if x_int is not None:
                return self._gemm(x_int)
            # the reference's two launches (nn/Linear.py:162-176) as one wherever the quantizing GEMM
            # takes the shape; otherwise literally: quantize, then GEMM
            if qlinear_f16in_wanted(x, N, K, w4=self.w_packed4):
                return self._gemm_f16in(x)
            return self._gemm(quant_op(x, self.act_scales_inv, self.act_zero_points))
        # BOS carve-out: token 0 is a precomputed FP16 row, tokens 1.. go through the kernels
        out = _bos_out
        if out is not None and not (out.shape == (x.shape[0], x.shape[1], N) and out.dtype == torch.float16
                                    and out.device == x.device and out.is_contiguous()):
            out = None
        if qlinear_f16in_wanted(x, N, K, w4=self.w_packed4, bos=True):
            if out is None:
                out = torch.empty((x.shape[0], x.shape[1], N), dtype=torch.float16, device=x.device)
                out[:, :1, :] = self.bos_pre_computed
            return self._gemm_f16in(x, out=out, bos=True)
        x_int = quant_op(x[:, 1:, :], self.act_scales_inv, self.act_zero_points)
        return self.forward_bos_quantized(x_int, x.shape[0], x.shape[1], out=out)
