"""`mixdq_extension._C` for MI355X: the five operator entry points of the reference's pybind
module (csrc/main.cpp:9-13), same names, argument order, kwarg names and error behaviour, backed
by libmixdq_hip.so (include/mixdq_hip.h) through ctypes.

    quantize_per_tensor_to_int8(input, scale_inv, zero_point)              quantize.cc:9-30
    quantize_per_tensor_to_int8_vectorized(input, scale_inv, zero_point)   quantize.cc:32-53
    qlinear_w8_a8_ohalf(input_int8, weight_int8, weight_scale, input_scale, input_zero_point,
                        weight_sum_by_input_channels, scale, bias0, bias=None)   qlinear.cc:13-137
    qconv2d_w8_a8_ohalf(input_int8, weight_int8, weight_scale, input_scale, input_zero_point,
                        scale, weight_sum_by_input_channels, bias0, bias=None, stride=1,
                        padding=0, dilation=1)                               qconv2d.cc:27-206
    qlinear_fp_reference(input, weight, bias=None)                           qlinear.cc:140-204

PyTorch is plumbing here: device memory (torch.empty -> caching allocator, so hipGraph capture
works), the current stream and error propagation.  There is NO CPU or eager fallback: if the HIP
library is missing the import fails, and non-GPU tensors raise like the reference's TORCH_CHECKs.
"""
from __future__ import annotations

import ctypes
import os

import torch

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MIXDQ_HIP_LIB") or os.path.join(_PKG, "libmixdq_hip.so")

if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} not found: the HIP extension has not been built. Run "
        "`python -m mixdq_amd.build` (needs hipcc). There is no CPU fallback.")

_lib = ctypes.CDLL(LIB_PATH)

_vp, _i64, _i32, _sz = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_size_t
_lib.mixdq_status_string.restype = ctypes.c_char_p
_lib.mixdq_status_string.argtypes = [_i32]
_lib.mixdq_abi_version.restype = _i32
_lib.mixdq_quantize_f16_i8.argtypes = [_vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _i32, _vp]
_lib.mixdq_qlinear_w8a8.argtypes = [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _vp]
_lib.mixdq_qlinear_w8a8_rows.argtypes = [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32,
                                         _i32, _i32, _i32, _vp, _i64, _i32, _vp]
_lib.mixdq_qconv2d_workspace_bytes.restype = _sz
_lib.mixdq_qconv2d_workspace_bytes.argtypes = [_i32] * 4
_lib.mixdq_qconv2d_w8a8.argtypes = [_vp] * 9 + [_i32] * 11 + [_vp]
_lib.mixdq_conv_border_table.argtypes = [_vp, _vp, _i32, _i32, _i32, _vp]
_lib.mixdq_qconv2d_w8a8_table.argtypes = [_vp] * 8 + [_i32] * 9 + [_vp, _i64, _i32, _vp]
_lib.mixdq_conv_zero_point_propagate.argtypes = [_vp, _vp, _vp] + [_i32] * 8 + [_vp]
_lib.mixdq_gemm_f16.argtypes = [_vp, _vp, _vp, _i64, _i32, _i32, _vp]
for _n in ("mixdq_quantize_f16_i8", "mixdq_qlinear_w8a8", "mixdq_qlinear_w8a8_rows",
           "mixdq_qconv2d_w8a8", "mixdq_conv_border_table", "mixdq_qconv2d_w8a8_table",
           "mixdq_conv_zero_point_propagate", "mixdq_gemm_f16"):
    getattr(_lib, _n).restype = _i32

_lib.mixdq_igemm_select.argtypes = [_i64, _i32, _i32, _i32, _vp, _vp, _vp, _vp]
_lib.mixdq_igemm_select.restype = _i32

ABI_VERSION = _lib.mixdq_abi_version()
# This binding is written against ONE data-layout contract (include/mixdq_hip.h MIXDQ_ABI_VERSION: e.g. ABI 3
# stores GEMM+GEGLU weight rows as value|gate groups of 16 where ABI 2 had 32): a library of another
# version (MIXDQ_HIP_LIB A/B runs) would compute silently different tensors -- refuse it here.
EXPECTED_ABI = 3
if ABI_VERSION != EXPECTED_ABI:
    raise ImportError(f"{LIB_PATH} reports ABI version {ABI_VERSION}; mixdq_amd._C is written for "
                      f"ABI {EXPECTED_ABI} (rebuild with `python -m mixdq_amd.build --force`)")


_lib.mixdq_igemm_select_id.argtypes = [_i64, _i32, _i32, _i32]
_lib.mixdq_igemm_select_id.restype = _i32
_lib.mixdq_igemm_select_id_w4.argtypes = [_i64, _i32, _i32, _i32]
_lib.mixdq_igemm_select_id_w4.restype = _i32


_lib.mixdq_igemm_select_id_geglu.argtypes = [_i64, _i32, _i32, _i32]
_lib.mixdq_igemm_select_id_geglu.restype = _i32


def igemm_select_id(M: int, N: int, k_align: int, k_total: int = 0, w4: bool = False,
                    geglu: bool = False) -> int:
    """Configuration id (IGEMM_CONFIGS key) the automatic choice makes for this problem
    (`w4`: for packed 4-bit weights, MIXDQ_FLAG_W4; `geglu`: for the GEMM + GEGLU launch)."""
    if geglu:
        return int(_lib.mixdq_igemm_select_id_geglu(M, N, k_total or k_align, int(w4)))
    if w4:
        return int(_lib.mixdq_igemm_select_id_w4(M, N, k_align, k_total or k_align))
    return int(_lib.mixdq_igemm_select_id(M, N, k_align, k_total or k_align))


def igemm_select(M: int, N: int, k_align: int, k_total: int = 0):
    """(BM, BN, BK, STAGES) of the igemm_kernel instantiation used for this problem; zeros =
    the generic small-alignment kernel."""
    bm, bn, bk, st = ctypes.c_int(), ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    _status(_lib.mixdq_igemm_select(M, N, k_align, k_total or k_align, ctypes.byref(bm),
                                    ctypes.byref(bn),
                                    ctypes.byref(bk), ctypes.byref(st)), "igemm_select")
    return bm.value, bn.value, bk.value, st.value

# Kernel configurations of the INT8 GEMM / conv family (csrc/igemm.hip MIXDQ_IGEMM_CONFIGS):
# id -> (BM, BN, BK, STAGES).  `_cfg=id` forces one (tuning and tests); 0 = automatic.
IGEMM_CONFIGS = {1: (64, 64, 64, 2), 3: (128, 128, 64, 2), 4: (64, 64, 128, 3),
                 13: (256, 128, 64, 3), 14: (256, 256, 64, 3), 15: (128, 256, 64, 3),
                 18: (256, 128, 128, 2), 20: (256, 256, 128, 2), 25: (128, 320, 128, 2),
                 27: (128, 320, 128, 2),   # the same tile on 16 waves of 16 x 160 (16x16x64 MFMAs)
                 28: (128, 320, 128, 2),   # ... on 16 waves of 32 x 80 (7 fragment reads per 10 MFMAs instead of 11)
                 35: (128, 128, 64, 3), 37: (64, 64, 128, 3), 41: (64, 128, 128, 3),
                 # 16x16x64-MFMA tiles (exactly one workgroup per CU on the UNet's M = 1024 / 4096
                 # layers; 45 / 56: deeper pipelines) and the 4-stage 128x320 tile
                 42: (64, 80, 128, 3), 43: (64, 240, 128, 3), 44: (128, 80, 128, 3),
                 45: (64, 80, 128, 4), 46: (128, 320, 64, 4), 56: (64, 80, 128, 6),
                 # 256x256x128 on the four-phase loop (2 x 4 waves of 128x64, 16x16x64 MFMAs)
                 70: (256, 256, 128, 2),
                 # ... and its persistent form: one workgroup per CU walking its tiles (csrc/igemm_pp.h); what the
                 # automatic choice takes instead of 70 where a CU has more than one tile
                 71: (256, 256, 128, 2)}

# id -> (WM, WN, KSPLIT, MT): wave grid, k-split groups and MFMA shape of each configuration (the template
# arguments that tell two configurations of one tile shape apart in a kernel trace)
IGEMM_WAVES = {1: (2, 2, 1, 32), 3: (2, 2, 1, 32), 4: (2, 2, 1, 32), 13: (4, 2, 1, 32), 14: (4, 2, 1, 32),
               15: (2, 4, 1, 32), 18: (4, 2, 1, 32), 20: (4, 2, 1, 32), 25: (4, 2, 1, 32), 27: (8, 2, 1, 16),
               28: (4, 4, 1, 16), 35: (4, 2, 1, 32), 37: (2, 2, 2, 32), 41: (2, 4, 1, 32), 42: (4, 1, 2, 16),
               43: (4, 1, 2, 16), 44: (4, 1, 2, 16), 45: (4, 1, 2, 16), 46: (4, 2, 1, 32), 47: (4, 2, 1, 32),
               56: (4, 1, 2, 16), 70: (2, 4, 1, 16), 71: (2, 4, 1, 16)}

FLAG_W4 = 2   # MIXDQ_FLAG_W4: the weight tensor holds packed signed 4-bit values
FLAG_UPSAMPLE2X = 4   # MIXDQ_FLAG_UPSAMPLE2X: the conv input is read through a nearest 2x upsampling

# Rounding variant of the fused multiply-adds: "A" (default) = FMA,
# "B" = separate multiply and add.  Read once at import; no other global state.
FLAGS = 1 if os.environ.get("MIXDQ_EPILOGUE_VARIANT", "A").upper() == "B" else 0


# Launch recorder (measurement only: bench.py's roofline leg, tools/).  While `RECORD` is a list,
# every INT8 GEMM / conv entry point appends (name, (M, N, K, k_align), w4, replay) to it, where
# replay() re-issues the very same launch on the same device tensors.  None = off (the default).
RECORD = None


def _record(name, M, N, K, k_align, w4, fn, args, kwargs):
    if RECORD is None:
        return
    def replay():
        global RECORD
        saved, RECORD = RECORD, None
        try:
            return fn(*args, **kwargs)
        finally:
            RECORD = saved
    RECORD.append((name, (int(M), int(N), int(K), int(k_align)), bool(w4), replay))


class PrefetchContext:
    """One forward's weight trace and prefetch plan (mixdq_amd/unet.py's planner).

    While a context is active on the calling THREAD (`with ctx:`), every GEMM / conv entry point appends the
    weight operand it was called with to `trace` and every long-key attention launch a marker -- the
    execution order of the network's weights relative to its self-attention launches -- and the i-th such
    attention launch carries `plan[i]` (weak references to weight tensors) as its payload.  The state lives
    in the object, never in the module: two UNets, or two threads each driving a device, keep their own.
    Only tensors on `device` are traced or handed to a launch (a payload address of another device would be
    dereferenced by the kernel).  A context does not nest: entering one while another is active on the
    thread leaves the outer one in charge and makes the inner one a no-op."""
    __slots__ = ("device", "trace", "plan", "att_index", "_outer", "_live")

    def __init__(self, device, plan=None):
        self.device = torch.device(device)
        self.trace = []
        self.plan = plan
        self.att_index = 0
        self._outer = None
        self._live = False

    def __enter__(self):
        self._outer = getattr(_TLS, "ctx", None)
        if self._outer is None:
            _TLS.ctx = self
            self._live = True
        return self

    def __exit__(self, *exc):
        if self._live:
            _TLS.ctx = None
            self._live = False
        return False

    def payload(self):
        """The tensors of the next long-key attention launch (dead references and other devices dropped)."""
        i, self.att_index = self.att_index, self.att_index + 1
        if self.plan is None or i >= len(self.plan):
            return None
        out = []
        for ref in self.plan[i]:
            t = ref()
            if t is not None and t.device == self.device:
                out.append(t)
        return out


import threading as _threading   # noqa: E402
import weakref as _weakref       # noqa: E402, F401  (plans hold weakref.ref(tensor))
_TLS = _threading.local()


def prefetch_context():
    """The PrefetchContext active on this thread, or None."""
    return getattr(_TLS, "ctx", None)


def _trace_w(t):
    ctx = getattr(_TLS, "ctx", None)
    if ctx is not None and t is not None and t.device == ctx.device:
        ctx.trace.append(t)


def _check(cond: bool, msg: str):
    if not cond:
        raise RuntimeError(msg)


def _status(code: int, what: str):
    if code != 0:
        raise RuntimeError(f"{what}: {_lib.mixdq_status_string(code).decode()}")


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _ptr(t):
    return None if t is None else t.data_ptr()


def _i64arr(vals):
    return (ctypes.c_int64 * len(vals))(*vals)


def _f32vec(t: torch.Tensor) -> torch.Tensor:
    """Per-channel epilogue vectors are read with 16-byte loads: contiguous and aligned."""
    t = t.contiguous()
    if t.data_ptr() % 16:
        t = t.clone()
    return t


def _quantize(input, scale_inv, zero_point):
    _check(input.is_cuda, "input should be on CUDA")
    _check(input.device == scale_inv.device, "input and scale should be on the same device")
    _check(input.device == zero_point.device,
           "input and zero_point should be on the same device")
    _check(input.dtype == torch.float16, "input should be fp16")
    _check(scale_inv.dtype == torch.float32, "scale_inv should be fp32")
    _check(zero_point.dtype == torch.float32, "zero_point should be fp32")
    out = torch.empty_like(input, dtype=torch.int8)
    if input.numel() == 0:
        return out
    nd = input.dim()
    sizes = list(input.shape) if nd else [1]
    xs = list(input.stride()) if nd else [1]
    os_ = list(out.stride()) if nd else [1]
    with torch.cuda.device(input.device):
        code = _lib.mixdq_quantize_f16_i8(input.data_ptr(), out.data_ptr(), _i64arr(sizes),
                                          _i64arr(xs), _i64arr(os_), len(sizes),
                                          scale_inv.data_ptr(), zero_point.data_ptr(), FLAGS,
                                          _stream())
    _status(code, "quantize_per_tensor_to_int8")
    return out


def quantize_per_tensor_to_int8(input, scale_inv, zero_point):
    """Quantize to int 8 per tensor with scale and zero point."""
    return _quantize(input, scale_inv, zero_point)


def quantize_per_tensor_to_int8_vectorized(input, scale_inv, zero_point):
    """Same kernel family as quantize_per_tensor_to_int8 (every path here is vectorised)."""
    return _quantize(input, scale_inv, zero_point)


def qlinear_w8_a8_ohalf(input_int8, weight_int8, weight_scale, input_scale, input_zero_point,
                        weight_sum_by_input_channels, scale, bias0, bias=None, *,
                        _out=None, _row_map=None, _cfg=0, _residual=None, _residual_div=1,
                        _w4=False):
    _trace_w(weight_int8)
    _check(input_int8.is_cuda, "Input should be on GPU.")
    dev = input_int8.device
    _check(dev == weight_int8.device, "input and weight_int8 should be on the same device.")
    _check(dev == weight_scale.device, "input and weight_scale should be on the same device.")
    _check(dev == input_scale.device, "input and input_scale should be on the same device.")
    _check(dev == input_zero_point.device,
           "input and input_zero_point should be on the same device.")
    _check(dev == weight_sum_by_input_channels.device,
           "input and input_zero_point should be on the same device.")
    if bias is not None:
        _check(dev == bias.device, "input and bias should be on the same device.")
    _check(input_int8.dtype == torch.int8, "input_int8 should be int8 type")
    _check(weight_int8.dtype == torch.int8, "weight_int8 should be int8 type")
    _check(weight_scale.dtype == torch.float32,
           "Currently only support weight_scale with float32 type")
    _check(input_scale.dtype == torch.float32,
           "Currently only support input_scale with float32 type")
    _check(input_zero_point.dtype == torch.float32,
           "Currently only support input_zero_point with float32 type")
    _check(weight_sum_by_input_channels.dtype == torch.float32,
           "Currently only support weight_sum_by_input_channels with float32 type")
    if bias is not None:
        _check(bias.dtype == torch.float16, "Currently only support bias with float16 type")
    _check(scale.dtype == torch.float32 and bias0.dtype == torch.float32,
           "scale and bias0 should be float32")
    N, K = weight_int8.size(0), weight_int8.size(1) * (2 if _w4 else 1)   # _w4: packed nibbles
    _check(weight_scale.numel() == N,
           "The size of the weight_scale vector should be equal to output_channels.")
    _check(weight_sum_by_input_channels.numel() == N,
           "The size of weight_sum_by_input_channels should equal output_channels.")
    _check(scale.numel() == N and bias0.numel() == N,
           "The size of scale and bias0 should be equal to output_channels.")
    if bias is not None:
        _check(bias.numel() == N,
               "The size of the bias vector should be equal to output_channels.")
    _check(input_int8.size(-1) == K,
           f"The last dimension of input and weight should match, got {input_int8.size(-1)} "
           f"and {K}.")
    a = input_int8.contiguous()
    w = weight_int8.contiguous()
    M = a.numel() // K if K else 0
    D = _out if _out is not None else torch.empty(
        list(input_int8.shape[:-1]) + [N], dtype=torch.float16, device=dev)
    rm = _row_map or (0, 0, 0)
    sc, b0 = _f32vec(scale), _f32vec(bias0)
    bs = None if bias is None else bias.contiguous()
    if _residual is not None:
        _check(_residual.dtype == torch.float16 and _residual.is_contiguous()
               and _residual.numel() == (M // _residual_div) * N,
               "residual should be contiguous fp16 of M / residual_div rows")
    _record("linear", M, N, K, K, _w4, qlinear_w8_a8_ohalf,
            (input_int8, weight_int8, weight_scale, input_scale, input_zero_point,
             weight_sum_by_input_channels, scale, bias0, bias),
            dict(_out=_out, _row_map=_row_map, _cfg=_cfg, _residual=_residual,
                 _residual_div=_residual_div, _w4=_w4))
    with torch.cuda.device(dev):
        code = _lib.mixdq_qlinear_w8a8_rows(a.data_ptr(), w.data_ptr(), b0.data_ptr(),
                                            sc.data_ptr(), _ptr(bs), D.data_ptr(), M, N, K,
                                            rm[0], rm[1], rm[2], _ptr(_residual), _residual_div,
                                            FLAGS | (_cfg << 8) | (FLAG_W4 if _w4 else 0),
                                            _stream())
    _status(code, "qlinear_w8_a8_ohalf")
    return D


_lib.mixdq_qlinear_f16in_w8a8.argtypes = [_vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32,
                                         _i32, _i32, _i32, _vp, _i64, _i32, _vp]
_lib.mixdq_qlinear_f16in_w8a8.restype = _i32
_lib.mixdq_qlinear_f16in_supported.argtypes = [_i64, _i32, _i32, _i64, _i64, _i32]
_lib.mixdq_qlinear_f16in_supported.restype = _i32
_lib.mixdq_qlinear_f16in_select_id.argtypes = [_i64, _i32, _i32, _i32]
_lib.mixdq_qlinear_f16in_select_id.restype = _i32
_lib.mixdq_qlinear_f16in_preferred.argtypes = [_i64, _i32, _i32, _i32]
_lib.mixdq_qlinear_f16in_preferred.restype = _i32

FLAG_A_ROWMAP = 8   # MIXDQ_FLAG_A_ROWMAP: the FP16 operand's rows follow the output row map

# Quantize-in-prologue (csrc/igemm_aq.hip): QuantizedLinear / 1x1 QuantizedConv2d hand their FP16 input
# straight to a GEMM that quantizes it in its staging path -- one launch where the reference has two
# (nn/Linear.py:162-176).  MIXDQ_F16IN=0 restores the reference's quantize launch (A/B runs, tests).
# "auto" (default): where the library's cost model expects the one launch to be cheaper
# (mixdq_qlinear_f16in_preferred: narrow layers -- the 1x1 shortcuts, K <= 640); "1": wherever supported.
F16IN = os.environ.get("MIXDQ_F16IN", "auto").lower()
F16IN = {"0": "0", "off": "0", "1": "1", "on": "1"}.get(F16IN, "auto")
# Tile configurations the quantizing family is built for (csrc/igemm_aq.hip MIXDQ_AQ_CONFIGS)
F16IN_CONFIGS = (4, 13, 27, 28, 35, 37, 41, 44, 45, 56)


def _rows_view(x: torch.Tensor, K: int):
    """(M, lda, leading shape) if `x` [..., K] can be read in place as M rows of K contiguous FP16 values
    a constant `lda` elements apart (a dense tensor, or a last-dimension slice of one), else None."""
    if x.dim() == 0 or x.size(-1) != K or (K > 1 and x.stride(-1) != 1):
        return None
    lead = list(x.shape[:-1])
    M = 1
    for d in lead:
        M *= d
    if M == 0:
        return None
    lda = None
    expect = None                      # stride the next-outer dimension must have
    for size, stride in zip(reversed(lead), reversed(x.stride()[:-1])):
        if size == 1:
            continue
        if lda is None:
            lda, expect = stride, stride * size
        elif stride != expect:
            return None
        else:
            expect = stride * size
    if lda is None:
        lda = K
    return M, lda, lead


def qlinear_f16in_supported(x: torch.Tensor, N: int, K: int, *, w4: bool = False, bos: bool = False) -> bool:
    """True if qlinear_f16in takes `x` (fp16 [..., K]; bos: [B, T, K] whose tokens 1.. are the operand)."""
    if not (x.is_cuda and x.dtype == torch.float16 and x.data_ptr() % 16 == 0):
        return False
    if bos:
        if x.dim() != 3 or not x.is_contiguous() or x.size(1) < 2:
            return False
        B, T = x.size(0), x.size(1)
        return bool(_lib.mixdq_qlinear_f16in_supported(B * (T - 1), N, K, K, B * T, int(w4)))
    v = _rows_view(x, K)
    if v is None:
        return False
    M, lda, _ = v
    return bool(_lib.mixdq_qlinear_f16in_supported(M, N, K, lda, M, int(w4)))


def qlinear_f16in_wanted(x: torch.Tensor, N: int, K: int, *, w4: bool = False, bos: bool = False) -> bool:
    """What the modules ask: fuse this layer's quantize into its GEMM?  Supported, and -- under the default
    MIXDQ_F16IN=auto -- expected to be cheaper than the two launches (mixdq_qlinear_f16in_preferred)."""
    if F16IN == "0" or not qlinear_f16in_supported(x, N, K, w4=w4, bos=bos):
        return False
    if F16IN == "1":
        return True
    M = x.size(0) * (x.size(1) - 1) if bos else x.numel() // K
    return bool(_lib.mixdq_qlinear_f16in_preferred(M, N, K, int(w4)))


def qlinear_f16in(input_f16, scale_inv, zero_point, weight_int8, scale, bias0, bias=None, *,
                  _out=None, _bos=False, _cfg=0, _residual=None, _residual_div=1, _w4=False, _trace=None):
    """quantize_per_tensor_to_int8(input, scale_inv, zero_point) -> qlinear_w8_a8_ohalf(...) in ONE launch
    (bit-identical to the pair).  `input_f16`: fp16 [..., K] readable in place as rows a constant stride
    apart; `_bos`: input is [B, T, K] and tokens 1.. are the operand, written to rows 1.. of `_out`
    [B, T, N] (QuantizedLinear's BOS path).  Raises where qlinear_f16in_supported() is False.
    `_trace`: the PERSISTENT tensor the weight operand is a view of (a 1x1 conv hands over a fresh
    permute / reshape view of its buffer on every call: the prefetch planner keeps weak references, and a
    reference to a temporary is dead on the next forward -- ADVICE r5)."""
    _trace_w(weight_int8 if _trace is None else _trace)
    _check(input_f16.is_cuda and input_f16.dtype == torch.float16, "input should be fp16 on GPU")
    dev = input_f16.device
    for t, nm in ((scale_inv, "scale_inv"), (zero_point, "zero_point"), (weight_int8, "weight_int8"),
                  (scale, "scale"), (bias0, "bias0")):
        _check(t.device == dev, f"input and {nm} should be on the same device.")
    _check(scale_inv.dtype == torch.float32 and zero_point.dtype == torch.float32,
           "scale_inv and zero_point should be fp32")
    _check(weight_int8.dtype == torch.int8, "weight_int8 should be int8 type")
    _check(scale.dtype == torch.float32 and bias0.dtype == torch.float32,
           "scale and bias0 should be float32")
    N, K = weight_int8.size(0), weight_int8.size(1) * (2 if _w4 else 1)
    _check(scale.numel() == N and bias0.numel() == N,
           "The size of scale and bias0 should be equal to output_channels.")
    if bias is not None:
        _check(bias.dtype == torch.float16 and bias.numel() == N and bias.device == dev,
               "bias should be fp16 of output_channels on the input's device")
    _check(input_f16.size(-1) == K,
           f"The last dimension of input and weight should match, got {input_f16.size(-1)} and {K}.")
    if _bos:
        _check(input_f16.dim() == 3 and input_f16.is_contiguous() and input_f16.size(1) >= 2,
               "BOS input should be a contiguous [B, T, K] tensor with T >= 2")
        B, T = input_f16.size(0), input_f16.size(1)
        M, lda, rm, flags = B * (T - 1), K, (T - 1, T, 1), FLAG_A_ROWMAP
        D = _out if _out is not None else torch.empty((B, T, N), dtype=torch.float16, device=dev)
    else:
        v = _rows_view(input_f16, K)
        _check(v is not None, "input rows should be a constant stride apart")
        M, lda, lead = v
        rm, flags = (0, 0, 0), 0
        D = _out if _out is not None else torch.empty(lead + [N], dtype=torch.float16, device=dev)
    w = weight_int8.contiguous()
    sc, b0 = _f32vec(scale), _f32vec(bias0)
    bs = None if bias is None else bias.contiguous()
    if _residual is not None:
        _check(_residual.dtype == torch.float16 and _residual.is_contiguous()
               and _residual.numel() == (M // _residual_div) * N,
               "residual should be contiguous fp16 of M / residual_div rows")
    _record("linear_f16in", M, N, K, K, _w4, qlinear_f16in,
            (input_f16, scale_inv, zero_point, weight_int8, scale, bias0, bias),
            dict(_out=_out, _bos=_bos, _cfg=_cfg, _residual=_residual, _residual_div=_residual_div,
                 _w4=_w4))
    with torch.cuda.device(dev):
        code = _lib.mixdq_qlinear_f16in_w8a8(
            input_f16.data_ptr(), lda, scale_inv.data_ptr(), zero_point.data_ptr(), w.data_ptr(),
            b0.data_ptr(), sc.data_ptr(), _ptr(bs), D.data_ptr(), M, N, K, rm[0], rm[1], rm[2],
            _ptr(_residual), _residual_div,
            FLAGS | flags | (_cfg << 8) | (FLAG_W4 if _w4 else 0), _stream())
    _status(code, "qlinear_f16in")
    return D


_lib.mixdq_qlinear_w8a8_grouped.argtypes = [_vp, _vp, _i32, _i64, _i32, _i32, _i32, _i32, _i32, _i32, _vp]
_lib.mixdq_qlinear_w8a8_grouped.restype = _i32


class GemmGroupTable:
    """Device array of `mixdq_gemm_group` members (include/mixdq_hip.h) for qlinear_grouped.
    members: iterable of (weight_int8 [N, K], bias0 [N], scale [N], bias fp16 [N] or None,
    out fp16 tensor whose data_ptr() is the member's output base).  The table keeps references to
    every tensor it points at; `key` identifies the storage it was built for."""

    def __init__(self, members, w4=False):
        import numpy as np
        self.members = [tuple(m) for m in members]
        self.w4 = bool(w4)
        rows, keep = [], []
        for w, b0, sc, bias, out in self.members:
            _check(w.is_cuda and w.dtype == torch.int8 and w.is_contiguous(), "member weight: int8")
            N = w.size(0)
            b0, sc = _f32vec(b0), _f32vec(sc)
            bias = None if bias is None else bias.contiguous()
            _check(b0.numel() == N and sc.numel() == N and (bias is None or bias.numel() == N),
                   "member epilogue vectors should have N elements")
            # the kernel cannot validate a device-side table: row stride == the member's N, whole
            # 8-byte output quads, 16-byte aligned operands
            _check(out.dtype == torch.float16 and out.data_ptr() % 16 == 0 and out.is_contiguous()
                   and out.dim() >= 1 and out.shape[-1] == N and N > 0 and out.numel() % N == 0,
                   "member output: contiguous fp16 [..., N], 16-byte aligned")
            _check(N % 4 == 0, "Int8 kernel with input or output alignment not to 4 is not supported.")
            _check(w.data_ptr() % 16 == 0 and (bias is None or bias.data_ptr() % 8 == 0)
                   and (bias is None or bias.dtype == torch.float16),
                   "member weight / bias: 16- / 8-byte aligned, bias fp16")
            keep.append((w, b0, sc, bias, out))
            rows.append([w.data_ptr(), b0.data_ptr(), sc.data_ptr(),
                         0 if bias is None else bias.data_ptr(), out.data_ptr(), N])
        self._keep = keep
        self.K = self.members[0][0].size(1) * (2 if w4 else 1)
        _check(all(m[0].size(1) == self.members[0][0].size(1) for m in self.members),
               "members of a grouped launch share K")
        self.max_N = max(r[5] for r in rows)
        self.n = len(rows)
        # struct mixdq_gemm_group: 5 pointers, int32 N, int32 reserved = 6 x 8 bytes
        self.table = torch.from_numpy(np.asarray(rows, dtype=np.int64)).to(self.members[0][0].device)
        self.key = tuple(r[0] for r in rows) + tuple(r[4] for r in rows)


def qlinear_grouped(input_int8, table: "GemmGroupTable", *, _row_map=None, _cfg=0):
    """`table.n` Linears on the same int8 input in one launch (mixdq_qlinear_w8a8_grouped); every
    member writes its own output tensor (the ones the table was built with)."""
    _check(input_int8.is_cuda and input_int8.dtype == torch.int8, "input_int8 should be int8 on GPU")
    _check(input_int8.size(-1) == table.K, "The last dimension of input and weight should match")
    a = input_int8.contiguous()
    M = a.numel() // table.K if table.K else 0
    rm = _row_map or (0, 0, 0)
    _record("linear_grouped", M, sum(m[0].size(0) for m in table.members), table.K, table.K,
            table.w4, qlinear_grouped,
            (input_int8, table), dict(_row_map=_row_map, _cfg=_cfg))
    with torch.cuda.device(a.device):
        code = _lib.mixdq_qlinear_w8a8_grouped(a.data_ptr(), table.table.data_ptr(), table.n, M,
                                               table.max_N, table.K, rm[0], rm[1], rm[2],
                                               FLAGS | (_cfg << 8) | (FLAG_W4 if table.w4 else 0),
                                               _stream())
    _status(code, "qlinear_grouped")


_lib.mixdq_qlinear_w8a8_attn.argtypes = [_vp] * 7 + [_i64, _i32, _i32, _i32, _i32, _i64, _i32, _i64,
                                          _i32, ctypes.c_float, _vp, _vp, _i32, _vp]
_lib.mixdq_qlinear_w8a8_attn.restype = _i32


def qlinear_attention_supported(x_shape, N, K, k) -> bool:
    """Shapes the fused to_q + cross-attention launch takes (see qlinear_attention)."""
    return (len(x_shape) == 3 and N % 128 == 0 and K % 128 == 0 and x_shape[1] % 64 == 0
            and k.dim() == 3 and 0 < k.shape[1] <= 128 and k.shape[2] == N
            and x_shape[0] * x_shape[1] * K < 2 ** 32 and N * K < 2 ** 32)


def qlinear_attention(input_int8, weight_int8, scale, bias0, k, v, scale_inv=None, zero_point=None,
                      softmax_scale=None, *, _w4=False):
    """attn2.to_q (INT8 GEMM, no bias) and the cross-attention core in one launch
    (mixdq_qlinear_w8a8_attn): input int8 [B, T, K], weight [N, K], k / v fp16 [B, Tkv <= 128, N]
    with unit stride along N.  Returns the attention output [B, T, N]: int8 (to_out.0's operand)
    when scale_inv / zero_point are given, else fp16.  Bit-identical to qlinear_w8_a8_ohalf
    followed by attention_f16."""
    _trace_w(weight_int8)
    _check(input_int8.is_cuda and input_int8.dtype == torch.int8 and input_int8.dim() == 3,
           "input_int8 should be an int8 [B, T, K] GPU tensor")
    _check(weight_int8.dtype == torch.int8, "weight_int8 should be int8 type")
    N, K = weight_int8.size(0), weight_int8.size(1) * (2 if _w4 else 1)
    B, T, _ = input_int8.shape
    _check(input_int8.size(-1) == K, "The last dimension of input and weight should match")
    for t, n in ((k, "k"), (v, "v")):
        _check(t.is_cuda and t.dtype == torch.float16 and t.dim() == 3 and t.stride(-1) == 1
               and t.shape[0] == B and t.shape[2] == N, f"{n} should be fp16 [B, Tkv, N]")
    _check(k.shape == v.shape, "k / v shapes disagree")
    _check(qlinear_attention_supported(input_int8.shape, N, K, k),
           "qlinear_attention: unsupported configuration (N % 128, K % 128, T % 64, Tkv <= 128)")
    quant = scale_inv is not None
    a, w = input_int8.contiguous(), weight_int8.contiguous()
    out = torch.empty((B, T, N), dtype=torch.int8 if quant else torch.float16, device=a.device)
    sc, b0 = _f32vec(scale), _f32vec(bias0)
    ss = float(softmax_scale) if softmax_scale is not None else 0.125
    _record("linear_attn", B * T, N, K, K, _w4, qlinear_attention,
            (input_int8, weight_int8, scale, bias0, k, v, scale_inv, zero_point, softmax_scale),
            dict(_w4=_w4))
    with torch.cuda.device(a.device):
        code = _lib.mixdq_qlinear_w8a8_attn(
            a.data_ptr(), w.data_ptr(), b0.data_ptr(), sc.data_ptr(), k.data_ptr(), v.data_ptr(),
            out.data_ptr(), B * T, N, K, T, k.shape[1], k.stride(0), k.stride(1), v.stride(0),
            v.stride(1), ss, _ptr(scale_inv), _ptr(zero_point),
            FLAGS | (FLAG_W4 if _w4 else 0), _stream())
    _status(code, "qlinear_attention")
    return out


_lib.mixdq_qlinear_w8a8_geglu.argtypes = [_vp] * 6 + [_i64, _i32, _i32, _vp, _vp, _i32, _vp]
_lib.mixdq_qlinear_w8a8_geglu.restype = _i32


GELU_TABLE_MAG = 0x4c00   # csrc/igemm.hip kGeluTabMag: the table covers |gate| < 16


def silu_table(device="cuda"):
    """(table [n_pos + n_neg] uint16 as int16 tensor, n_pos, n_neg): the FP16 -> FP16 SiLU table of the GroupNorm apply
    pass's large-launch variant (mixdq_silu_table); building it eagerly keeps the init kernel out of captured graphs."""
    _lib.mixdq_silu_table.argtypes = [_vp, _vp, _vp, _vp]
    _lib.mixdq_silu_table.restype = _i32
    n_pos, n_neg = ctypes.c_int(0), ctypes.c_int(0)
    with torch.cuda.device(device):
        _status(_lib.mixdq_silu_table(None, ctypes.byref(n_pos), ctypes.byref(n_neg), _stream()), "silu_table")
        out = torch.empty(n_pos.value + n_neg.value, dtype=torch.int16, device=device)
        _status(_lib.mixdq_silu_table(out.data_ptr(), None, None, _stream()), "silu_table")
    return out, n_pos.value, n_neg.value


def gelu_table(device="cuda") -> torch.Tensor:
    """The table the GEMM + GEGLU epilogue of the large tiles looks GELU up in (mixdq_gelu_table):
    int16 [2, 0x4c00] -- row 0: gates +0 .. +16, row 1: -0 .. -16 -- of f16 bit patterns."""
    _lib.mixdq_gelu_table.argtypes = [_vp, _vp]
    _lib.mixdq_gelu_table.restype = _i32
    out = torch.empty((2, GELU_TABLE_MAG), dtype=torch.int16, device=device)
    with torch.cuda.device(out.device):
        _status(_lib.mixdq_gelu_table(out.data_ptr(), _stream()), "gelu_table")
    return out


def geglu_row_order(D: int, device=None) -> torch.Tensor:
    """Row order of mixdq_qlinear_w8a8_geglu's weight: value|gate groups of 16.  perm[i] = the row
    of the ordinary [2D, K] GEGLU projection (values 0..D-1, gates D..2D-1) stored at row i."""
    g = torch.arange(D // 16, device=device)[:, None] * 16 + torch.arange(16, device=device)[None, :]
    return torch.stack([g, g + D], dim=1).reshape(-1)


def qlinear_geglu(input_int8, weight_int8, scale, bias0, bias, out_scale_inv, out_zero_point, *,
                  _cfg=0, _w4=False, _out=None):
    """int8 [..., K] x value/gate-interleaved W [2D, K] -> int8 [..., D]: ff.net.0.proj + GEGLU +
    the quantizer of ff.net.2 in one launch (include/mixdq_hip.h: mixdq_qlinear_w8a8_geglu).
    `_out`: a contiguous int8 tensor of the result's size to write into (8-byte aligned)."""
    _trace_w(weight_int8)
    _check(input_int8.is_cuda and input_int8.dtype == torch.int8, "input_int8 should be int8 on GPU")
    _check(weight_int8.dtype == torch.int8, "weight_int8 should be int8 type")
    N, K = weight_int8.size(0), weight_int8.size(1) * (2 if _w4 else 1)
    _check(input_int8.size(-1) == K, "The last dimension of input and weight should match")
    _check(scale.numel() == N and bias0.numel() == N, "scale and bias0 should have 2D elements")
    a, w = input_int8.contiguous(), weight_int8.contiguous()
    M = a.numel() // K if K else 0
    out = _out if _out is not None else torch.empty(list(input_int8.shape[:-1]) + [N // 2],
                                                    dtype=torch.int8, device=a.device)
    _check(out.dtype == torch.int8 and out.is_contiguous() and out.numel() == M * (N // 2),
           "_out should be a contiguous int8 tensor of the result's size")
    sc, b0 = _f32vec(scale), _f32vec(bias0)
    bs = None if bias is None else bias.contiguous()
    _record("linear_geglu", M, N, K, K, _w4, qlinear_geglu,
            (input_int8, weight_int8, scale, bias0, bias, out_scale_inv, out_zero_point),
            dict(_cfg=_cfg, _w4=_w4))
    with torch.cuda.device(a.device):
        code = _lib.mixdq_qlinear_w8a8_geglu(a.data_ptr(), w.data_ptr(), b0.data_ptr(),
                                             sc.data_ptr(), _ptr(bs), out.data_ptr(), M, N, K,
                                             _ptr(out_scale_inv), _ptr(out_zero_point),
                                             FLAGS | (_cfg << 8) | (FLAG_W4 if _w4 else 0),
                                             _stream())
    _status(code, "qlinear_geglu")
    return out


if hasattr(_lib, "mixdq_conv_halo_select"):      # (absent in older builds used for A/B runs)
    _lib.mixdq_conv_halo_select.argtypes = [_i32] * 9
    _lib.mixdq_conv_halo_select.restype = _i32
HALO_TILES = {90: (8, 16, 80), 91: (8, 8, 80), 92: (16, 16, 80), 93: (16, 16, 160)}   # csrc/iconv.hip: output pixels (rows, columns), channels


def conv_halo_select(N, H, W, C, K, R, S, stride, padding) -> int:
    """Tile id (HALO_TILES key) of the LDS-resident-halo kernel an unforced INT8 conv of this shape
    runs on, or 0 (the implicit-GEMM family)."""
    if not hasattr(_lib, "mixdq_conv_halo_select"):
        return 0
    return int(_lib.mixdq_conv_halo_select(N, H, W, C, K, R, S, stride, padding))


def _conv_geometry(input_int8, weight_int8, stride, padding, dilation):
    N, C, H, W = input_int8.shape
    K, _, R, S = weight_int8.shape
    P = (H + 2 * padding - dilation * (R - 1) - 1) // stride + 1
    Q = (W + 2 * padding - dilation * (S - 1) - 1) // stride + 1
    return N, C, H, W, K, R, S, P, Q


def conv_upsample2x_supported(x_shape, weight_shape, stride, padding) -> bool:
    """qconv2d_w8_a8_ohalf(..., _upsample2x=True) takes this conv (the LDS-halo kernel's range)."""
    N, C, H, W = x_shape
    K, _, R, S = weight_shape
    return conv_halo_select(N, 2 * H, 2 * W, C, K, R, S, stride, padding) != 0


def qconv2d_w8_a8_ohalf(input_int8, weight_int8, weight_scale, input_scale, input_zero_point,
                        scale, weight_sum_by_input_channels, bias0, bias=None, stride=1,
                        padding=0, dilation=1, *, _table=None, _cfg=0, _residual=None,
                        _residual_per_image=False, _w4=False, _upsample2x=False):
    _trace_w(weight_int8)
    stride = 1 if stride is None else int(stride)
    padding = 0 if padding is None else int(padding)
    dilation = 1 if dilation is None else int(dilation)
    _check(input_int8.is_cuda, "Input should be on GPU.")
    dev = input_int8.device
    _check(dev == weight_int8.device, "input and weight_int8 should be on the same device.")
    _check(dev == weight_scale.device, "input and weight_scale should be on the same device.")
    _check(dev == input_scale.device, "input and input_scale should be on the same device.")
    _check(dev == input_zero_point.device,
           "input and input_zero_point should be on the same device.")
    wsum = weight_sum_by_input_channels
    if wsum is not None:
        _check(dev == wsum.device,
               "input and weight_sum_by_input_channels should be on the same device.")
    if bias0 is not None:
        _check(dev == bias0.device, "input and bias0 should be on the same device.")
    if bias is not None:
        _check(dev == bias.device, "input and bias should be on the same device.")
    _check(input_int8.dtype == torch.int8, "input_int8 should be int8 type")
    _check(weight_int8.dtype == torch.int8, "weight_int8 should be int8 type")
    _check(weight_scale.dtype == torch.float32,
           "Currently only support weight_scale with float32 type")
    _check(input_scale.dtype == torch.float32,
           "Currently only support input_scale with float32 type")
    _check(input_zero_point.dtype == torch.float32,
           "Currently only support input_zero_point with float32 type")
    if wsum is not None:
        _check(wsum.dtype == torch.float32,
               "Currently only support weight_sum_by_input_channels with float32 type")
    if bias0 is not None:
        _check(bias0.dtype == torch.float32, "Currently only support bias0 with float32 type")
    if bias is not None:
        _check(bias.dtype == torch.float16, "Currently only support bias with float16 type")
    _check(scale.dtype == torch.float32, "scale should be float32")
    _check(input_int8.dim() == 4 and weight_int8.dim() == 4, "input and weight should be 4-D")
    N, C, H, W, K, R, S, P, Q = _conv_geometry(input_int8, weight_int8, stride, padding, dilation)
    if _upsample2x:       # the conv runs on the nearest 2x upsampling of the stored input
        H, W = 2 * H, 2 * W
        P = (H + 2 * padding - (R - 1) - 1) // stride + 1
        Q = (W + 2 * padding - (S - 1) - 1) // stride + 1
    _check(weight_int8.size(1) * (2 if _w4 else 1) == C,
           "input and weight channel counts should match")
    _check(weight_scale.numel() == K,
           "The size of the weight_scale vector should be equal to output_channels.")
    if padding == 0:
        _check(bias0 is not None and bias0.numel() == K,
               "The size of bias0 should equal output_channels.")
    else:
        _check(wsum is not None and wsum.numel() == K * R * S,
               "The size of weight_sum_by_input_channels should equal K*R*S.")
    if bias is not None:
        _check(bias.numel() == K,
               "The size of the bias vector should be equal to output_channels.")
    x = input_int8.contiguous(memory_format=torch.channels_last)
    w = weight_int8.contiguous(memory_format=torch.channels_last)
    D = torch.empty((N, K, P, Q), dtype=torch.float16, device=dev,
                    memory_format=torch.channels_last)
    sc = _f32vec(scale)
    bs = None if bias is None else bias.contiguous()
    kind = "conv"
    if RECORD is not None and _cfg in (0, 90, 91, 92, 93) and not _w4 and dilation == 1:
        tile = _cfg or conv_halo_select(N, H, W, C, K, R, S, stride, padding)
        kind = f"conv_halo{tile}" if tile else "conv"
    _record(kind, N * P * Q, K, R * S * C, C, _w4, qconv2d_w8_a8_ohalf,
            (input_int8, weight_int8, weight_scale, input_scale, input_zero_point, scale,
             weight_sum_by_input_channels, bias0, bias, stride, padding, dilation),
            dict(_table=_table, _cfg=_cfg, _residual=_residual,
                 _residual_per_image=_residual_per_image, _w4=_w4, _upsample2x=_upsample2x))
    with torch.cuda.device(dev):
        res_ptr, res_div = None, 1
        if _residual is not None:
            _check(_residual.dtype == torch.float16, "residual should be fp16")
            if _residual_per_image:       # [N, K] (or [N, K, 1, 1]) row per image
                _check(_residual.is_contiguous() and _residual.numel() == N * K,
                       "per-image residual should be contiguous [N, K]")
                res_div = P * Q
            else:                         # [N, K, P, Q] channels-last
                _check(tuple(_residual.shape) == (N, K, P, Q) and _residual.is_contiguous(
                    memory_format=torch.channels_last), "residual should be channels-last [N,K,P,Q]")
            res_ptr = _residual.data_ptr()
        if padding > 0 and _table is None and _residual is None and not _upsample2x:
            ws_bytes = _lib.mixdq_qconv2d_workspace_bytes(K, R, S, padding)
            workspace = torch.empty(ws_bytes // 4, dtype=torch.float32, device=dev)
            code = _lib.mixdq_qconv2d_w8a8(
                x.data_ptr(), w.data_ptr(), sc.data_ptr(), wsum.contiguous().data_ptr(),
                input_zero_point.data_ptr(), None, _ptr(bs), D.data_ptr(), workspace.data_ptr(),
                N, H, W, C, K, R, S, stride, padding, dilation,
                FLAGS | (_cfg << 8) | (FLAG_W4 if _w4 else 0), _stream())
        else:
            _check(dilation == 1, "qconv2d_w8_a8_ohalf: unsupported configuration "
                                  "(dilation must be 1)")
            if padding > 0 and _table is None:
                _table = conv_border_table(wsum)
            b0 = None if padding > 0 else _f32vec(bias0)
            code = _lib.mixdq_qconv2d_w8a8_table(
                x.data_ptr(), w.data_ptr(), sc.data_ptr(), _ptr(_table),
                input_zero_point.data_ptr(), _ptr(b0), _ptr(bs), D.data_ptr(),
                N, H, W, C, K, R, S, stride, padding, res_ptr, res_div,
                FLAGS | (_cfg << 8) | (FLAG_W4 if _w4 else 0)
                | (FLAG_UPSAMPLE2X if _upsample2x else 0), _stream())
    _status(code, "qconv2d_w8_a8_ohalf")
    return D


def conv_border_table(weight_sum_by_input_channels, out=None):
    """Tap-rectangle sums of wsum [K,1,R,S] -> [(R*R*S*S), K] f32 (include/mixdq_hip.h,
    mixdq_conv_border_table).  Depends only on the weights: QuantizedConv2d caches it (`out`: an
    existing table to rebuild in place)."""
    wsum = weight_sum_by_input_channels.contiguous()
    _check(wsum.is_cuda and wsum.dtype == torch.float32 and wsum.dim() == 4,
           "weight_sum_by_input_channels should be a float32 [K,1,R,S] GPU tensor")
    K, _, R, S = wsum.shape
    table = out
    if table is None:
        table = torch.empty((R * R * S * S, K), dtype=torch.float32, device=wsum.device)
    _check(tuple(table.shape) == (R * R * S * S, K) and table.dtype == torch.float32
           and table.is_contiguous() and table.device == wsum.device, "border table shape")
    with torch.cuda.device(wsum.device):
        code = _lib.mixdq_conv_border_table(wsum.data_ptr(), table.data_ptr(), K, R, S, _stream())
    _status(code, "conv_border_table")
    return table


def conv_zero_point_propagate(weight_sum_by_input_channels, input_zero_point, N, H, W, stride,
                              padding):
    """The reference's materialised bias0 [N,K,P,Q] f32 channels-last
    (conv_act_zero_point_propagate.cu:54-83); diagnostic, not on the fast path."""
    wsum = weight_sum_by_input_channels.contiguous()
    K, _, R, S = wsum.shape
    P = (H + 2 * padding - (R - 1) - 1) // stride + 1
    Q = (W + 2 * padding - (S - 1) - 1) // stride + 1
    out = torch.empty((N, K, P, Q), dtype=torch.float32, device=wsum.device,
                      memory_format=torch.channels_last)
    with torch.cuda.device(wsum.device):
        code = _lib.mixdq_conv_zero_point_propagate(
            wsum.data_ptr(), input_zero_point.data_ptr(), out.data_ptr(), N, H, W, K, R, S,
            stride, padding, _stream())
    _status(code, "conv_zero_point_propagate")
    return out


def qlinear_fp_reference(input, weight, bias=None):
    """FP16 debug GEMM: input [..., K] @ weight [K, N] (row-major, qlinear.cc:161).  `bias` is
    accepted and ignored, as in the reference (op/qlinear.py:92)."""
    _check(input.dtype == torch.float16, "input should be int8 type")      # sic, qlinear.cc:146
    _check(weight.dtype == torch.float16, "weight should be int8 type")    # sic, qlinear.cc:148
    _check(input.is_cuda and weight.is_cuda, "Input should be on GPU.")
    if bias is not None:
        _check(input.device == bias.device, "input and bias should be on the same device.")
        _check(bias.dtype == torch.float16, "Currently only support bias with float16 type")
    K, N = weight.size(0), weight.size(1)
    a = input.contiguous()
    b = weight.contiguous()
    M = a.numel() // K if K else 0
    D = torch.empty(list(input.shape[:-1]) + [N], dtype=torch.float16, device=input.device)
    with torch.cuda.device(input.device):
        code = _lib.mixdq_gemm_f16(a.data_ptr(), b.data_ptr(), D.data_ptr(), M, N, K, _stream())
    _status(code, "qlinear_fp_reference")
    return D


# ---------------------------------------------------------------------------------------------
# FP16 layers (include/mixdq_hip.h: mixdq_linear_f16 / mixdq_conv2d_f16): the reference's FP fallback
# (F.linear / F.conv2d on the FP16 weight) on this repo's own MFMA kernels.
# ---------------------------------------------------------------------------------------------
_lib.mixdq_linear_f16.argtypes = [_vp, _vp, _vp, _vp, _i64, _i32, _i32, _vp, _i64, _i32, _vp]
_lib.mixdq_linear_f16.restype = _i32
_lib.mixdq_conv2d_f16.argtypes = [_vp] * 4 + [_i32] * 9 + [_vp, _i64, _i32, _vp]
_lib.mixdq_conv2d_f16.restype = _i32


def linear_f16(input, weight, bias=None, *, _residual=None, _cfg=0):
    """F.linear(input, weight, bias) for fp16 GPU tensors: input [..., K], weight [N, K] -> [..., N],
    FP32 accumulation, bias added in FP32, one rounding to fp16; `_residual` (fp16, the output's
    shape) is added after that rounding as a following torch half add would."""
    _trace_w(weight)
    _check(input.is_cuda and input.dtype == torch.float16, "input should be an fp16 GPU tensor")
    _check(weight.dtype == torch.float16 and weight.device == input.device and weight.dim() == 2,
           "weight should be an fp16 [N, K] tensor on the input's device")
    N, K = weight.shape
    _check(input.size(-1) == K, "The last dimension of input and weight should match")
    if bias is not None:
        _check(bias.dtype == torch.float16 and bias.numel() == N, "bias should be fp16 [N]")
    a, w = input.contiguous(), weight.contiguous()
    M = a.numel() // K if K else 0
    D = torch.empty(list(input.shape[:-1]) + [N], dtype=torch.float16, device=a.device)
    if _residual is not None:
        _check(_residual.dtype == torch.float16 and _residual.is_contiguous()
               and _residual.numel() == M * N, "residual should be contiguous fp16 [M, N]")
    bs = None if bias is None else bias.contiguous()
    with torch.cuda.device(a.device):
        code = _lib.mixdq_linear_f16(a.data_ptr(), w.data_ptr(), _ptr(bs), D.data_ptr(), M, N, K,
                                     _ptr(_residual), 1, int(_cfg) << 8, _stream())
    _status(code, "linear_f16")
    return D


def conv2d_f16(input, weight, bias=None, stride=1, padding=0, *, _residual=None,
               _residual_per_image=False, _cfg=0):
    """F.conv2d(input, weight, bias, stride, padding) for fp16 GPU tensors (square stride / padding,
    dilation 1, groups 1): input [N, C, H, W] and weight [K, C, R, S] are read in channels-last
    memory (converted if they are not), the result is channels-last [N, K, P, Q]."""
    _trace_w(weight)
    _check(input.is_cuda and input.dtype == torch.float16 and input.dim() == 4,
           "input should be a 4-D fp16 GPU tensor")
    _check(weight.dtype == torch.float16 and weight.device == input.device and weight.dim() == 4,
           "weight should be a 4-D fp16 tensor on the input's device")
    stride, padding = int(stride), int(padding)
    N, C, H, W, K, R, S, P, Q = _conv_geometry(input, weight, stride, padding, 1)
    _check(weight.size(1) == C, "input and weight channel counts should match")
    if bias is not None:
        _check(bias.dtype == torch.float16 and bias.numel() == K, "bias should be fp16 [K]")
    x = input.contiguous(memory_format=torch.channels_last)
    w = weight.contiguous(memory_format=torch.channels_last)
    D = torch.empty((N, K, P, Q), dtype=torch.float16, device=x.device,
                    memory_format=torch.channels_last)
    res_ptr, res_div = None, 1
    if _residual is not None:
        _check(_residual.dtype == torch.float16, "residual should be fp16")
        if _residual_per_image:
            _check(_residual.is_contiguous() and _residual.numel() == N * K,
                   "per-image residual should be contiguous [N, K]")
            res_div = P * Q
        else:
            _check(tuple(_residual.shape) == (N, K, P, Q) and _residual.is_contiguous(
                memory_format=torch.channels_last), "residual should be channels-last [N,K,P,Q]")
        res_ptr = _residual.data_ptr()
    bs = None if bias is None else bias.contiguous()
    with torch.cuda.device(x.device):
        code = _lib.mixdq_conv2d_f16(x.data_ptr(), w.data_ptr(), _ptr(bs), D.data_ptr(), N, H, W, C,
                                     K, R, S, stride, padding, res_ptr, res_div, int(_cfg) << 8,
                                     _stream())
    _status(code, "conv2d_f16")
    return D


F16_CONFIGS = (4, 13, 20, 25, 35, 41)   # csrc/igemm.hip MIXDQ_F16_CONFIGS (one accumulation order)


# ---------------------------------------------------------------------------------------------
# Producer fusions (include/mixdq_hip.h, csrc/fused_norm.hip).  Not part of the reference's `_C`:
# used by mixdq_amd.unet's fused forward.  Same no-fallback rule: GPU tensors only.
# ---------------------------------------------------------------------------------------------
_lib.mixdq_groupnorm_workspace_bytes.restype = _sz
_lib.mixdq_groupnorm_workspace_bytes.argtypes = [_i32, _i64, _i32, _i32]
_lib.mixdq_groupnorm_silu_quantize.restype = _i32
_lib.mixdq_groupnorm_silu_quantize.argtypes = [_vp, _vp, _vp, ctypes.c_float, _i32, _vp, _vp, _vp,
                                               _vp, _vp, _i32, _i64, _i32, _i32, _i32, _vp]
_lib.mixdq_groupnorm_silu_quantize2.restype = _i32
_lib.mixdq_groupnorm_silu_quantize2.argtypes = [_vp, _i32, _vp, _vp, _vp, ctypes.c_float, _i32, _vp,
                                                _vp, _vp, _vp, _vp, _i32, _i64, _i32, _i32, _i32, _vp]
_lib.mixdq_groupnorm_silu_quantize3.restype = _i32
_lib.mixdq_groupnorm_silu_quantize3.argtypes = [_vp, _i32, _vp, _vp, _vp, ctypes.c_float, _i32, _vp,
                                                _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i64, _i32,
                                                _i32, _i32, _vp]
_lib.mixdq_layernorm_quantize.restype = _i32
_lib.mixdq_layernorm_quantize.argtypes = [_vp, _vp, _vp, ctypes.c_float, _i64, _i32, _i32, _vp, _vp,
                                          _vp, _vp, _i32, _vp]
_lib.mixdq_geglu_quantize.restype = _i32
_lib.mixdq_geglu_quantize.argtypes = [_vp, _i64, _i32, _vp, _vp, _vp, _vp, _i32, _vp]


def groupnorm_supported(N, HW, C, G) -> bool:
    return _lib.mixdq_groupnorm_workspace_bytes(N, HW, C, G) > 0


def groupnorm_silu_quantize(x, num_groups, weight, bias, eps, scale_inv=None, zero_point=None,
                            silu=True, want_f16=False, x2=None, raw_qparams=None):
    """x: fp16 [N, C, H, W] in channels-last memory (or [N, HW, C] contiguous).  Returns
    (int8 or None, fp16 or None) with x's shape and strides.
    x2 (same layout, same N / H / W): the GroupNorm of cat([x, x2], dim=channels) without making
    the concatenation; the outputs have the concatenated shape.
    raw_qparams: one (scale_inv, zero_point) pair or None per source tensor; the return value gains
    a third element, the list of `quantize_per_tensor(source)` results (None where not asked for),
    written by the same pass."""
    _check(x.is_cuda and x.dtype == torch.float16, "x should be an fp16 GPU tensor")
    if x.dim() == 4:
        _check(x.is_contiguous(memory_format=torch.channels_last),
               "groupnorm_silu_quantize needs channels-last input")
        N, C, HW = x.shape[0], x.shape[1], x.shape[2] * x.shape[3]
    else:
        _check(x.dim() == 3 and x.is_contiguous(), "x should be [N, HW, C] contiguous")
        N, HW, C = x.shape
    C1 = C
    shape = tuple(x.shape)
    if x2 is not None:
        _check(x2.is_cuda and x2.dtype == torch.float16 and x2.dim() == x.dim()
               and x2.shape[0] == N and (x2.shape[2:] == x.shape[2:] if x.dim() == 4
                                         else x2.shape[1] == HW)
               and (x2.is_contiguous(memory_format=torch.channels_last) if x.dim() == 4
                    else x2.is_contiguous()), "x2 should match x in batch / spatial size and layout")
        C2 = x2.shape[1] if x.dim() == 4 else x2.shape[2]
        _check(C1 % 8 == 0 and C2 % 8 == 0, "two-source GroupNorm needs channel counts % 8 == 0")
        C = C1 + C2
        shape = (N, C, x.shape[2], x.shape[3]) if x.dim() == 4 else (N, HW, C)
    fmt = dict(memory_format=torch.channels_last) if x.dim() == 4 else {}
    want_q = scale_inv is not None
    _check(want_q or want_f16, "nothing to compute")
    ws_bytes = _lib.mixdq_groupnorm_workspace_bytes(N, HW, C, num_groups)
    _check(ws_bytes > 0, "groupnorm_silu_quantize: unsupported configuration")
    ws = torch.empty(ws_bytes // 4, dtype=torch.float32, device=x.device)
    out_q = torch.empty(shape, dtype=torch.int8, device=x.device, **fmt) if want_q else None
    out_h = torch.empty(shape, dtype=torch.float16, device=x.device, **fmt) if want_f16 else None
    w, b = weight.contiguous(), bias.contiguous()
    _check(w.dtype == torch.float16 and b.dtype == torch.float16, "gamma/beta should be fp16")
    raws = None
    arr = ctypes.c_void_p * 2
    r_si = r_zp = r_q = None
    if raw_qparams is not None:
        sources = [x] if x2 is None else [x, x2]
        _check(len(raw_qparams) == len(sources), "one raw quantizer slot per source tensor")
        raws = [None if qp is None else torch.empty_like(src, dtype=torch.int8)
                for qp, src in zip(raw_qparams, sources)]
        pad = [None] * (2 - len(sources))
        r_si = arr(*[_ptr(None if qp is None else qp[0]) for qp in list(raw_qparams) + pad])
        r_zp = arr(*[_ptr(None if qp is None else qp[1]) for qp in list(raw_qparams) + pad])
        r_q = arr(*[_ptr(t) for t in raws + pad])
    with torch.cuda.device(x.device):
        code = _lib.mixdq_groupnorm_silu_quantize3(
            x.data_ptr(), C1, _ptr(x2), w.data_ptr(), b.data_ptr(), float(eps),
            int(bool(silu)), _ptr(scale_inv), _ptr(zero_point), _ptr(out_q), _ptr(out_h),
            r_si, r_zp, r_q, ws.data_ptr(), N, HW, C, num_groups, FLAGS, _stream())
    _status(code, "groupnorm_silu_quantize")
    if raw_qparams is not None:
        return out_q, out_h, raws
    return out_q, out_h


def layernorm_quantize(x, weight, bias, eps, qparams, want_f16=False):
    """x: fp16 [..., C] contiguous; qparams: up to three (scale_inv, zero_point) device-scalar
    pairs.  Returns ([int8 ...], fp16 or None)."""
    _check(x.is_cuda and x.dtype == torch.float16 and x.is_contiguous(),
           "x should be a contiguous fp16 GPU tensor")
    C = x.shape[-1]
    M = x.numel() // C if C else 0
    n = len(qparams)
    _check(n <= 3 and (n > 0 or want_f16), "layernorm_quantize: 1..3 quantizers or want_f16")
    outs = [torch.empty_like(x, dtype=torch.int8) for _ in range(n)]
    out_h = torch.empty_like(x) if want_f16 else None
    arr = ctypes.c_void_p * max(n, 1)
    si = arr(*[p[0].data_ptr() for p in qparams])
    zp = arr(*[p[1].data_ptr() for p in qparams])
    oq = arr(*[o.data_ptr() for o in outs])
    w, b = weight.contiguous(), bias.contiguous()
    _check(w.dtype == torch.float16 and b.dtype == torch.float16, "gamma/beta should be fp16")
    with torch.cuda.device(x.device):
        code = _lib.mixdq_layernorm_quantize(x.data_ptr(), w.data_ptr(), b.data_ptr(), float(eps),
                                             M, C, n, si, zp, oq, _ptr(out_h), FLAGS, _stream())
    _status(code, "layernorm_quantize")
    return outs, out_h


_lib.mixdq_qlinear_ln_workspace_bytes.restype = _sz
_lib.mixdq_qlinear_ln_workspace_bytes.argtypes = [_i64, _i32]
_lib.mixdq_qlinear_ln_select_id.restype = _i32
_lib.mixdq_qlinear_ln_select_id.argtypes = [_i64, _i32, _i32]
_lib.mixdq_qlinear_w8a8_ln.restype = _i32
_lib.mixdq_qlinear_w8a8_ln.argtypes = [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _vp, _i64, _vp, _vp,
                                       ctypes.c_float, _i32, _vp, _vp, _vp, _vp, _vp, _i32, _vp]

# GEMM + residual + LayerNorm + quantize in one launch (csrc/igemm_ln.hip).  MIXDQ_LN_FUSE=0: off (A/B runs).
LN_FUSE = os.environ.get("MIXDQ_LN_FUSE", "1") != "0"


def qlinear_ln_supported(M: int, N: int, K: int) -> bool:
    return bool(LN_FUSE and _lib.mixdq_qlinear_ln_select_id(int(M), int(N), int(K)) > 0)


def qlinear_ln_workspace(M: int, N: int, device) -> torch.Tensor:
    """A zeroed exchange buffer for qlinear_ln launches of up to M rows x N columns: its first page holds the
    launch counter (the epoch: it only grows -- every launch tags its records with epoch + 1), the departure
    counter (zero between launches) and a sticky error word (qlinear_ln_status); the records follow.  Owned by
    the caller: one launch at a time per buffer AND per device (the tiles of a launch wait for each other)."""
    return torch.zeros(int(_lib.mixdq_qlinear_ln_workspace_bytes(int(M), int(N))), dtype=torch.uint8,
                       device=device)


_lib.mixdq_qlinear_ln_status.restype = _i32
_lib.mixdq_qlinear_ln_status.argtypes = [_vp, _vp, _vp]


def qlinear_ln_status(workspace) -> int:
    """The workspace's sticky error word: 0 while every launch on it found all its records, else the tag of the
    last launch in which a workgroup gave up waiting (its rows were written as NaN).  Synchronises the stream."""
    w = ctypes.c_int(0)
    with torch.cuda.device(workspace.device):
        code = _lib.mixdq_qlinear_ln_status(workspace.data_ptr(), ctypes.byref(w), _stream())
    _status(code, "qlinear_ln_status")
    return int(w.value)


def qlinear_ln(input_int8, weight_int8, scale, bias0, bias, residual, ln_weight, ln_bias, eps, qparams,
               workspace, *, want_f16=False, _cfg=0):
    """y = qlinear_w8_a8_ohalf(input_int8, ...) [+ residual];  layernorm_quantize(y, ln_weight, ln_bias, eps,
    qparams, want_f16) -- in ONE launch, bit-identical to the two (mixdq_qlinear_w8a8_ln).  Returns
    (y fp16 [..., N], [int8 ...], fp16 or None).  Raises where qlinear_ln_supported() is False."""
    _trace_w(weight_int8)
    _check(input_int8.is_cuda and input_int8.dtype == torch.int8, "input_int8 should be int8 on GPU")
    dev = input_int8.device
    N, K = weight_int8.size(0), weight_int8.size(1)
    _check(input_int8.size(-1) == K, "The last dimension of input and weight should match")
    a, w = input_int8.contiguous(), weight_int8.contiguous()
    M = a.numel() // K if K else 0
    lead = list(input_int8.shape[:-1])
    D = torch.empty(lead + [N], dtype=torch.float16, device=dev)
    n = len(qparams)
    _check(n <= 3 and (n > 0 or want_f16), "qlinear_ln: 1..3 quantizers or want_f16")
    outs = [torch.empty(lead + [N], dtype=torch.int8, device=dev) for _ in range(n)]
    out_h = torch.empty(lead + [N], dtype=torch.float16, device=dev) if want_f16 else None
    arr = ctypes.c_void_p * max(n, 1)
    si = arr(*[p[0].data_ptr() for p in qparams])
    zp = arr(*[p[1].data_ptr() for p in qparams])
    oq = arr(*[o.data_ptr() for o in outs])
    sc, b0 = _f32vec(scale), _f32vec(bias0)
    bs = None if bias is None else bias.contiguous()
    g, b = ln_weight.contiguous(), ln_bias.contiguous()
    _check(g.dtype == torch.float16 and b.dtype == torch.float16 and g.numel() == N and b.numel() == N,
           "LayerNorm weight / bias should be fp16 [N]")
    if residual is not None:
        _check(residual.dtype == torch.float16 and residual.is_contiguous() and residual.numel() == M * N,
               "residual should be contiguous fp16 [M, N]")
    _check(workspace.is_cuda and workspace.device == dev and workspace.numel() >=
           _lib.mixdq_qlinear_ln_workspace_bytes(M, N), "workspace too small (qlinear_ln_workspace)")
    _record("linear_ln", M, N, K, K, False, qlinear_ln,
            (input_int8, weight_int8, scale, bias0, bias, residual, ln_weight, ln_bias, eps, qparams, workspace),
            dict(want_f16=want_f16, _cfg=_cfg))
    with torch.cuda.device(dev):
        code = _lib.mixdq_qlinear_w8a8_ln(a.data_ptr(), w.data_ptr(), b0.data_ptr(), sc.data_ptr(), _ptr(bs),
                                          D.data_ptr(), M, N, K, _ptr(residual), 1, g.data_ptr(), b.data_ptr(),
                                          float(eps), n, si, zp, oq, _ptr(out_h), workspace.data_ptr(),
                                          FLAGS | (_cfg << 8), _stream())
    _status(code, "qlinear_ln")
    return D, outs, out_h


def geglu_quantize(h, scale_inv=None, zero_point=None, want_f16=False):
    """h: fp16 [..., 2D] contiguous -> (int8 [..., D] or None, fp16 [..., D] or None)."""
    _check(h.is_cuda and h.dtype == torch.float16 and h.is_contiguous(),
           "h should be a contiguous fp16 GPU tensor")
    D = h.shape[-1] // 2
    M = h.numel() // (2 * D) if D else 0
    want_q = scale_inv is not None
    _check(want_q or want_f16, "nothing to compute")
    shape = list(h.shape[:-1]) + [D]
    out_q = torch.empty(shape, dtype=torch.int8, device=h.device) if want_q else None
    out_h = torch.empty(shape, dtype=torch.float16, device=h.device) if want_f16 else None
    with torch.cuda.device(h.device):
        code = _lib.mixdq_geglu_quantize(h.data_ptr(), M, D, _ptr(scale_inv), _ptr(zero_point),
                                         _ptr(out_q), _ptr(out_h), FLAGS, _stream())
    _status(code, "geglu_quantize")
    return out_q, out_h


_lib.mixdq_attention_f16.argtypes = [_vp] * 4 + [_i32] * 5 + [_i64] * 8 + [ctypes.c_float, _vp, _vp,
                                                                       _i32, _vp]
_lib.mixdq_attention_f16.restype = _i32


if hasattr(_lib, "mixdq_attention_f16_prefetch"):     # (absent in older builds used for A/B runs)
    _lib.mixdq_attention_f16_prefetch.argtypes = [_vp] * 4 + [_i32] * 5 + [_i64] * 8 + [
        ctypes.c_float, _vp, _vp, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_int64), _i32, _i32, _vp]
    _lib.mixdq_attention_f16_prefetch.restype = _i32
PREFETCH_MAX_RANGES = 16


def attention_f16(q, k, v, heads, scale_inv=None, zero_point=None, softmax_scale=None, _cfg=0, _prefetch=None):
    """FP16 attention core (the reference's get_attention_scores + bmm, quant_block.py:630-637).

    q [B, Tq, C], k/v [B, Tkv, C] fp16 with unit stride along C (column slices of a fused projection
    are read in place); C = heads * 64.  Returns fp16 [B, Tq, C], or — when `scale_inv`/`zero_point`
    (to_out.0's activation quantizer) are given — its int8 quantization.
    `_prefetch`: up to 16 GPU tensors (the weights of the layers behind this attention) that payload
    workgroups of the launch read while the attention runs (mixdq_attention_f16_prefetch); no effect
    on the result.
    """
    for t, n in ((q, "q"), (k, "k"), (v, "v")):
        _check(t.is_cuda and t.dtype == torch.float16 and t.dim() == 3 and t.stride(-1) == 1,
               f"{n} should be a [B, T, C] fp16 GPU tensor with unit stride along C")
    B, Tq, C = q.shape
    _check(k.shape == v.shape and k.shape[0] == B and k.shape[2] == C, "q/k/v shapes disagree")
    _check(C == heads * 64, "head_dim must be 64")
    quant = scale_inv is not None
    # (measurement only) recorded as ("attention", (B * heads * Tq, Tkv, 64, 64)): 4 * M * N * K FLOPs
    _record("attention", B * heads * Tq, k.shape[1], 64, 64, False, attention_f16,
            (q, k, v, heads), dict(scale_inv=scale_inv, zero_point=zero_point,
                                   softmax_scale=softmax_scale, _cfg=_cfg))     # (measured without the payload)
    out = torch.empty((B, Tq, C), dtype=torch.int8 if quant else torch.float16, device=q.device)
    sc = float(softmax_scale) if softmax_scale is not None else 0.125
    ctx = getattr(_TLS, "ctx", None)
    if k.shape[1] > 128 and ctx is not None and ctx.device == q.device:   # a launch that can carry a payload
        ctx.trace.append(("attn", B * Tq, k.shape[1]))
        planned = ctx.payload()
        if _prefetch is None:
            _prefetch = planned
    # (a payload address of another device would be dereferenced by this device's kernel: dropped)
    pf = [t for t in (_prefetch or ()) if t is not None and t.device == q.device and t.numel() > 0]
    _check(len(pf) <= PREFETCH_MAX_RANGES, "at most 16 prefetch ranges")
    with torch.cuda.device(q.device):
        if pf and hasattr(_lib, "mixdq_attention_f16_prefetch"):
            ptrs = (ctypes.c_void_p * len(pf))(*[t.data_ptr() for t in pf])
            sizes = (ctypes.c_int64 * len(pf))(*[t.numel() * t.element_size() for t in pf])
            code = _lib.mixdq_attention_f16_prefetch(
                q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), B, heads, 64, Tq, k.shape[1],
                q.stride(0), q.stride(1), k.stride(0), k.stride(1), v.stride(0), v.stride(1),
                out.stride(0), out.stride(1), sc, _ptr(scale_inv), _ptr(zero_point), ptrs, sizes, len(pf),
                FLAGS | (int(_cfg) << 8), _stream())
        else:
            code = _lib.mixdq_attention_f16(
                q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), B, heads, 64, Tq, k.shape[1],
                q.stride(0), q.stride(1), k.stride(0), k.stride(1), v.stride(0), v.stride(1),
                out.stride(0), out.stride(1), sc, _ptr(scale_inv), _ptr(zero_point),
                FLAGS | (int(_cfg) << 8), _stream())
    _status(code, "attention_f16")
    return out
