"""ctypes front-end of oracle/mixdq_oracle.c (+ a NumPy mirror used to cross-check the C).

TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py.  The arithmetic specification and the
reference file:line each function follows are in mixdq_oracle.c; "parity pinning" is described
in that file's header and in DESIGN.md.

All functions take/return NumPy arrays.  FP16 tensors are np.float16; the C side sees their
bit patterns as uint16.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libmixdq_oracle.so")
_lib = None

VARIANT_FUSED = 0    # variant A: fmaf (default; nvcc -fmad=true contraction)
VARIANT_UNFUSED = 1  # variant B: separate mul and add


def build(force: bool = False) -> str:
    """Compile the C oracle with the committed Makefile (gcc only)."""
    if force or not os.path.exists(_LIB_PATH) or (
            os.path.getmtime(_LIB_PATH) < os.path.getmtime(os.path.join(_HERE, "mixdq_oracle.c"))):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B" if force else "all"])
    return _LIB_PATH


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = ctypes.CDLL(_LIB_PATH)
        vp, i64, i32, f32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_float
        L.mixdq_oracle_quantize.argtypes = [vp, vp, vp, vp, vp, i32, f32, f32, i32]
        L.mixdq_oracle_quantize.restype = None
        L.mixdq_oracle_qlinear.argtypes = [vp, vp, vp, vp, vp, vp, vp, i64, i32, i32, i32]
        L.mixdq_oracle_qlinear.restype = None
        L.mixdq_oracle_zp_propagate.argtypes = [vp, f32, vp] + [i32] * 10
        L.mixdq_oracle_zp_propagate.restype = None
        L.mixdq_oracle_qconv2d.argtypes = [vp, vp, vp, vp, f32, vp, vp, vp, vp] + [i32] * 10
        L.mixdq_oracle_qconv2d.restype = None
        L.mixdq_oracle_add_f16.argtypes = [vp, vp, vp, i64]
        L.mixdq_oracle_add_f16.restype = None
        L.mixdq_oracle_gemm_f16.argtypes = [vp, vp, vp, i64, i32, i32]
        L.mixdq_oracle_gemm_f16.restype = None
        L.mixdq_oracle_groupnorm_silu_quantize.argtypes = [vp, vp, vp, f32, i32, f32, f32, vp, vp,
                                                            i32, i64, i32, i32, i32]
        L.mixdq_oracle_groupnorm_silu_quantize.restype = i32
        L.mixdq_oracle_layernorm_quantize.argtypes = [vp, vp, vp, f32, i64, i32, i32, vp, vp, vp,
                                                      vp, i32]
        L.mixdq_oracle_layernorm_quantize.restype = None
        L.mixdq_oracle_geglu_quantize.argtypes = [vp, i64, i32, f32, f32, vp, vp, i32]
        L.mixdq_oracle_geglu_quantize.restype = None
        for fn in ("expf", "erff", "siluf", "geluf"):
            getattr(L, "mixdq_oracle_" + fn).argtypes = [f32]
            getattr(L, "mixdq_oracle_" + fn).restype = f32
        L.mixdq_oracle_h2f.argtypes = [ctypes.c_uint16]
        L.mixdq_oracle_h2f.restype = f32
        L.mixdq_oracle_f2h.argtypes = [f32]
        L.mixdq_oracle_f2h.restype = ctypes.c_uint16
        _lib = L
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def _c(a, dtype):
    return None if a is None else np.ascontiguousarray(a, dtype=dtype)


def _elem_strides(a: np.ndarray):
    return np.asarray([s // a.itemsize for s in a.strides], dtype=np.int64)


def quantize(x: np.ndarray, scale_inv: float, zero_point: float,
             variant: int = VARIANT_FUSED) -> np.ndarray:
    """a1.  x: float16 array of any strides.  Returns int8 of the same shape (C-contiguous)."""
    assert x.dtype == np.float16
    out = np.empty(x.shape, dtype=np.int8)
    if x.size == 0:
        return out
    sizes = np.asarray(x.shape if x.ndim else (1,), dtype=np.int64)
    xs = _elem_strides(x) if x.ndim else np.asarray([1], dtype=np.int64)
    os_ = _elem_strides(out) if x.ndim else np.asarray([1], dtype=np.int64)
    assert (xs >= 0).all(), "negative strides not supported"
    lib().mixdq_oracle_quantize(_p(x), _p(out), _p(sizes), _p(xs), _p(os_), int(sizes.size),
                                float(np.float32(scale_inv)), float(np.float32(zero_point)),
                                variant)
    return out


def qlinear(a_i8, w_i8, bias0, scale, bias=None, variant=VARIANT_FUSED, return_acc=False):
    """a2.  a_i8 [..., K] int8, w_i8 [N, K] int8, bias0/scale [N] f32, bias [N] f16 or None.
    Returns float16 [..., N] (and the exact int32 accumulators if return_acc)."""
    a = _c(a_i8, np.int8)
    w = _c(w_i8, np.int8)
    N, K = w.shape
    assert a.shape[-1] == K
    M = a.size // K if K else 0
    b0 = _c(np.asarray(bias0).reshape(-1), np.float32)
    sc = _c(np.asarray(scale).reshape(-1), np.float32)
    bs = None if bias is None else _c(np.asarray(bias).reshape(-1), np.float16)
    assert b0.size == N and sc.size == N and (bs is None or bs.size == N)
    D = np.empty(a.shape[:-1] + (N,), dtype=np.float16)
    acc = np.empty(a.shape[:-1] + (N,), dtype=np.int32) if return_acc else None
    if M:
        lib().mixdq_oracle_qlinear(_p(a), _p(w), _p(b0), _p(sc), _p(bs), _p(D), _p(acc),
                                   M, N, K, variant)
    return (D, acc) if return_acc else D


def conv_out_hw(H, W, R, S, stride, pad):
    return (H + 2 * pad - (R - 1) - 1) // stride + 1, (W + 2 * pad - (S - 1) - 1) // stride + 1


def zp_propagate(wsum_krs, zp, N, H, W, stride, pad):
    """a4.  wsum [K,1,R,S] or [K,R,S] f32 -> bias0 [N,P,Q,K] f32 (NHWC order)."""
    ws = _c(wsum_krs, np.float32)
    if ws.ndim == 4:
        ws = ws.reshape(ws.shape[0], ws.shape[2], ws.shape[3])
    K, R, S = ws.shape
    P, Q = conv_out_hw(H, W, R, S, stride, pad)
    out = np.empty((N, P, Q, K), dtype=np.float32)
    lib().mixdq_oracle_zp_propagate(_p(ws), float(np.float32(zp)), _p(out), N, H, W, K, R, S,
                                    P, Q, stride, pad)
    return out


def qconv2d(x_nhwc, w_krsc, scale, wsum=None, zp=0.0, bias0=None, bias=None, stride=1, pad=0,
            variant=VARIANT_FUSED, return_acc=False):
    """a3.  x [N,H,W,C] int8 (NHWC order), w [K,R,S,C] int8.  Returns float16 [N,P,Q,K]."""
    x = _c(x_nhwc, np.int8)
    w = _c(w_krsc, np.int8)
    N, H, W, C = x.shape
    K, R, S, C2 = w.shape
    assert C == C2
    sc = _c(np.asarray(scale).reshape(-1), np.float32)
    ws = None
    if pad > 0:
        assert wsum is not None
        ws = _c(np.asarray(wsum).reshape(K, R, S), np.float32)
    else:
        assert bias0 is not None
    b0 = None if bias0 is None else _c(np.asarray(bias0).reshape(-1), np.float32)
    bs = None if bias is None else _c(np.asarray(bias).reshape(-1), np.float16)
    P, Q = conv_out_hw(H, W, R, S, stride, pad)
    D = np.empty((N, P, Q, K), dtype=np.float16)
    acc = np.empty((N, P, Q, K), dtype=np.int32) if return_acc else None
    lib().mixdq_oracle_qconv2d(_p(x), _p(w), _p(sc), _p(ws), float(np.float32(zp)), _p(b0),
                               _p(bs), _p(D), _p(acc), N, H, W, C, K, R, S, stride, pad, variant)
    return (D, acc) if return_acc else D


def add_f16(a, b):
    a = _c(a, np.float16)
    b = _c(b, np.float16)
    out = np.empty_like(a)
    lib().mixdq_oracle_add_f16(_p(a), _p(b), _p(out), a.size)
    return out


def gemm_f16(a, b_kn):
    a = _c(a, np.float16)
    b = _c(b_kn, np.float16)
    K, N = b.shape
    M = a.size // K
    D = np.empty(a.shape[:-1] + (N,), dtype=np.float16)
    lib().mixdq_oracle_gemm_f16(_p(a), _p(b), _p(D), M, N, K)
    return D


def groupnorm_silu_quantize(x_nhwc, gamma, beta, eps, num_groups, silu, scale_inv, zero_point,
                            variant=VARIANT_FUSED):
    """Fused GroupNorm(+SiLU)+quantize on x [N, HW, C] (or [N,H,W,C]) float16.
    Returns (int8, float16) of x's shape."""
    x = _c(x_nhwc, np.float16)
    N, C = x.shape[0], x.shape[-1]
    HW = x.size // (N * C)
    g, b = _c(gamma, np.float16), _c(beta, np.float16)
    q = np.empty(x.shape, np.int8)
    h = np.empty(x.shape, np.float16)
    ok = lib().mixdq_oracle_groupnorm_silu_quantize(
        _p(x), _p(g), _p(b), float(np.float32(eps)), int(bool(silu)),
        float(np.float32(scale_inv)), float(np.float32(zero_point)), _p(q), _p(h), N, HW, C,
        num_groups, variant)
    if not ok:
        raise ValueError("unsupported GroupNorm geometry")
    return q, h


def layernorm_quantize(x, gamma, beta, eps, qparams, variant=VARIANT_FUSED):
    """Fused LayerNorm+quantize on x [..., C] float16; qparams = [(scale_inv, zp), ...] (<= 3).
    Returns ([int8, ...], float16)."""
    x = _c(x, np.float16)
    C = x.shape[-1]
    M = x.size // C
    g, b = _c(gamma, np.float16), _c(beta, np.float16)
    outs = [np.empty(x.shape, np.int8) for _ in qparams]
    h = np.empty(x.shape, np.float16)
    si = np.asarray([p[0] for p in qparams], dtype=np.float32)
    zp = np.asarray([p[1] for p in qparams], dtype=np.float32)
    ptrs = (ctypes.c_void_p * max(len(outs), 1))(*[o.ctypes.data for o in outs])
    lib().mixdq_oracle_layernorm_quantize(_p(x), _p(g), _p(b), float(np.float32(eps)), M, C,
                                          len(outs), _p(si), _p(zp), ptrs, _p(h), variant)
    return outs, h


def geglu_quantize(h, scale_inv, zero_point, variant=VARIANT_FUSED):
    """Fused GEGLU+quantize on h [..., 2D] float16 -> (int8 [..., D], float16 [..., D])."""
    h = _c(h, np.float16)
    D = h.shape[-1] // 2
    M = h.size // (2 * D)
    q = np.empty(h.shape[:-1] + (D,), np.int8)
    o = np.empty(h.shape[:-1] + (D,), np.float16)
    lib().mixdq_oracle_geglu_quantize(_p(h), M, D, float(np.float32(scale_inv)),
                                      float(np.float32(zero_point)), _p(q), _p(o), variant)
    return q, o


def attention_f16(q, k, v, heads, softmax_scale=None):
    """FP16 attention core, restated in float64: out[b, t, h*D + d] = softmax_k(q.k * scale) v per
    head (the reference's get_attention_scores + torch.bmm, quant_block.py:630-637, diffusers'
    AttnProcessor semantics; FP16 tensors, no mask).  q [B, Tq, C], k/v [B, Tkv, C] float16.
    Returns (float16 result, float64 result).  A floating-point op: the HIP kernel is held to a
    stated tolerance against this, not to bit equality (its INT8 output variant is then checked
    bit-exactly as quantize() of its own FP16 output)."""
    q = np.asarray(q, np.float16).astype(np.float64)
    k = np.asarray(k, np.float16).astype(np.float64)
    v = np.asarray(v, np.float16).astype(np.float64)
    B, Tq, C = q.shape
    D = C // heads
    sc = (1.0 / np.sqrt(D)) if softmax_scale is None else float(softmax_scale)
    qh = q.reshape(B, Tq, heads, D).transpose(0, 2, 1, 3)
    kh = k.reshape(B, -1, heads, D).transpose(0, 2, 1, 3)
    vh = v.reshape(B, -1, heads, D).transpose(0, 2, 1, 3)
    s = np.einsum("bhqd,bhkd->bhqk", qh, kh) * sc
    s -= s.max(axis=-1, keepdims=True)
    p = np.exp(s)
    p /= p.sum(axis=-1, keepdims=True)
    o = np.einsum("bhqk,bhkd->bhqd", p, vh).transpose(0, 2, 1, 3).reshape(B, Tq, C)
    return o.astype(np.float16), o


def unpack_w4(packed: np.ndarray) -> np.ndarray:
    """W4 storage of include/mixdq_hip.h (MIXDQ_FLAG_W4), restated independently: within each
    group of 8 k-values, byte j holds k[j] in its high nibble and k[4+j] in its low nibble, two's
    complement.  int8/uint8 [..., K/2] -> int8 [..., K].  The W4 kernels must equal the W8
    restatement run on these unpacked integers."""
    b = np.ascontiguousarray(packed).view(np.uint8).astype(np.int16)
    b = b.reshape(b.shape[:-1] + (b.shape[-1] // 4, 4))
    out = np.empty(b.shape[:-1] + (8,), dtype=np.int16)
    out[..., 0:4] = (b >> 4) & 0xF
    out[..., 4:8] = b & 0xF
    out = np.where(out >= 8, out - 16, out)
    return out.astype(np.int8).reshape(packed.shape[:-1] + (packed.shape[-1] * 2,))


# ------------------------------------------------------------------------------------------
# NumPy mirror: an independent second restatement of the same arithmetic, used by
# tests/test_oracle.py to cross-check the C code (two restatements that agree bit-for-bit).
# ------------------------------------------------------------------------------------------
def np_quantize(x, scale_inv, zero_point, variant=VARIANT_FUSED):
    xf = x.astype(np.float32)
    s = np.float32(scale_inv)
    z = np.float32(zero_point)
    if variant == VARIANT_FUSED:
        # exact product in float64 (24b x 24b fits 53b), one rounding of the sum to f32
        t = (xf.astype(np.float64) * np.float64(s) + np.float64(z)).astype(np.float32)
        # double rounding hazard: f64 sum may itself be inexact; guard by checking exactness
        # (|x*s| and z differ by < 2^29 in exponent for every input used in tests).
    else:
        t = (xf * s).astype(np.float32) + z
    r = np.rint(t)  # half-to-even
    return np.clip(r, -128, 127).astype(np.int8)


def np_epilogue(acc_i32, bias0, scale, bias=None, variant=VARIANT_FUSED):
    v = acc_i32.astype(np.float32)
    d = (v - bias0.astype(np.float32)).astype(np.float32)
    if bias is None:
        r = (d * scale.astype(np.float32)).astype(np.float32)
    elif variant == VARIANT_FUSED:
        r = (d.astype(np.float64) * scale.astype(np.float64)
             + bias.astype(np.float64)).astype(np.float32)
    else:
        r = (d * scale.astype(np.float32)).astype(np.float32) + bias.astype(np.float32)
    with np.errstate(over="ignore"):
        return r.astype(np.float16)


def np_qlinear(a_i8, w_i8, bias0, scale, bias=None, variant=VARIANT_FUSED):
    acc = a_i8.astype(np.int32).reshape(-1, a_i8.shape[-1]) @ w_i8.astype(np.int32).T
    out = np_epilogue(acc, np.asarray(bias0)[None, :], np.asarray(scale)[None, :],
                      None if bias is None else np.asarray(bias)[None, :], variant)
    return out.reshape(a_i8.shape[:-1] + (w_i8.shape[0],))
