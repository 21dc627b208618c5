"""Quantization-parameter plumbing for the quantized modules.

Counterpart of mixdq_extension/nn/utils.py: QParam (:64-67), create_qparams_from_dtype (:79-135),
get_quant_para (:412-458) and the uint4 pack helpers (:13-53).  The checkpoint layout read here is
the one kernels/convert_ckpt.py:17-46 writes ("new_ckpt.pth").
"""
from __future__ import annotations

import math
from collections import namedtuple

import torch

dtype_to_bw = {
    torch.qint8: 8,
    torch.quint8: 8,
    torch.quint4x2: 4,
    torch.quint2x4: 2,
    torch.float16: 16,
}


class QParam(namedtuple("QParam", ["qscheme", "dtype", "scales", "zero_points", "axis"],
                        defaults=[torch.per_tensor_affine, torch.quint8, 1.0, 0.0, 0])):
    @property
    def zp_float(self):
        return self.scales * self.zero_points


def bit_index(n_bit: int) -> int:
    """delta_list / zero_point_list are stacked for bit-widths [2, 4, 8] (sdxl_turbo.yaml:7):
    index = log2(bits) - 1 (nn/utils.py:415)."""
    return int(math.log2(n_bit) - 1)


def get_quant_para(ckpt, n_bit, module_name, quant_type, split=0, device=None):
    """Look one layer's scales / zero points up in the converted checkpoint.

    Returns (scales, zero_point, scales_0, zero_point_0); the `_0` pair is the second half of a
    split conv_shortcut (None when split == 0).  Activation zero points are stored in the uint8
    convention and shifted by -128 to int8 (nn/utils.py:428,452,457)."""
    idx = bit_index(n_bit)
    if quant_type not in ("weight", "act"):
        raise ValueError(f"quant_type must be 'weight' or 'act', got {quant_type!r}")
    shift = 128 if quant_type == "act" else 0

    def fetch(key):
        if key not in ckpt:
            raise AssertionError(f"{key} not in checkpoint")
        entry = ckpt[key]
        return entry["delta_list"][idx].to(device), (entry["zero_point_list"][idx] - shift).to(device)

    key = f"{module_name}.{quant_type}_quantizer"
    scales, zero_point = fetch(key)
    if split == 0:
        return scales, zero_point, None, None
    scales_0, zero_point_0 = fetch(key + "_0")
    return scales, zero_point, scales_0, zero_point_0


def create_qparams_from_dtype(dtype, device, is_channel_wise=False, num_kernels=None, ckpt=None,
                              module_name=None, bit_width=0, quant_type=None, split=0):
    """(QParam, QParam_0) for one tensor; (None, None) for an un-quantized (fp16) one."""
    if dtype == torch.float16:
        return None, None
    if dtype not in (torch.qint8, torch.quint8, torch.quint4x2):
        raise ValueError(f"Unsupported quantize dtype {dtype}")
    scales, zps, scales_0, zps_0 = get_quant_para(ckpt, bit_width, module_name, quant_type,
                                                  split=split, device=device)
    if is_channel_wise:
        assert num_kernels is not None
        scheme = torch.per_channel_affine
    else:
        scheme = torch.per_tensor_affine
    qparam = QParam(qscheme=scheme, scales=scales, zero_points=zps, dtype=dtype, axis=0)
    qparam_0 = None
    if split > 0:
        qparam_0 = QParam(qscheme=scheme, scales=scales_0, zero_points=zps_0, dtype=dtype, axis=0)
    return qparam, qparam_0


# ---- 4-bit storage helpers (nn/utils.py:13-53): high nibble = even index --------------------
def _bcast(v, ndim):
    return v.view(-1, *([1] * (ndim - 1)))


def quantize_per_tensor_uint4(input: torch.Tensor, scale, zero_point):
    scale_inv = 1.0 / _bcast(scale, input.dim())
    q = torch.clamp(torch.round(input * scale_inv) + _bcast(zero_point, input.dim()), 0, 15
                    ).to(torch.uint8)
    if input.dim() >= 4:
        assert input.shape[1] % 2 == 0
        return q[:, ::2, ...] << 4 | q[:, 1::2, ...]
    assert input.shape[-1] % 2 == 0
    return q[..., ::2] << 4 | q[..., 1::2]


def unpack_uint4(input):
    hi = (input >> 4).to(torch.uint8)
    lo = (input & 0b1111).to(torch.uint8)
    if input.dim() >= 4:
        shape = (input.shape[0], input.shape[1] * 2, *input.shape[2:])
        return torch.stack([hi, lo], dim=2).view(shape)
    shape = (*input.shape[:-1], input.shape[-1] * 2)
    return torch.stack([hi, lo], dim=-1).view(shape)


def dequantize_per_tensor_uint4(input, scale, zero_point):
    x = unpack_uint4(input)
    return (x.to(torch.float32) - _bcast(zero_point, x.dim())) * _bcast(scale, x.dim())


# ---- W4 storage of this build ("nibble-planar per 8", include/mixdq_hip.h MIXDQ_FLAG_W4) ------
# The reference defines a 4-bit pack (above: high nibble = even index) but no kernel reads it.
# The MI355X kernel wants an unpack of 3 VALU ops per packed dword, which this layout gives:
# within every group of 8 consecutive k, byte j holds k[j] in the high and k[4+j] in the low
# nibble (two's complement).  Last dimension is K (Linear [N, K]; Conv [K, R, S, C] i.e. the
# channels-last weight): packed last dimension = K / 2.
def pack_w4(q: torch.Tensor) -> torch.Tensor:
    """int8 values in [-8, 7], last dim % 8 == 0  ->  int8 packed, last dim / 2."""
    assert q.dtype == torch.int8 and q.shape[-1] % 8 == 0
    assert int(q.min()) >= -8 and int(q.max()) <= 7, "W4 values must be in [-8, 7]"
    g = q.reshape(*q.shape[:-1], q.shape[-1] // 8, 2, 4).to(torch.int16)   # [.., group, half, j]
    packed = ((g[..., 0, :] & 0xF) << 4) | (g[..., 1, :] & 0xF)
    return packed.to(torch.uint8).view(torch.int8).reshape(*q.shape[:-1], q.shape[-1] // 2)


def unpack_w4(packed: torch.Tensor) -> torch.Tensor:
    """Inverse of pack_w4: int8 packed [..., K/2] -> int8 values [..., K]."""
    b = packed.view(torch.uint8).to(torch.int16).reshape(*packed.shape[:-1], packed.shape[-1] // 4, 4)
    hi, lo = (b >> 4) & 0xF, b & 0xF
    v = torch.stack([hi, lo], dim=-2)                         # [.., group, half, j]
    v = torch.where(v >= 8, v - 16, v)
    return v.to(torch.int8).reshape(*packed.shape[:-1], packed.shape[-1] * 2)
