"""The C-ABI library loads on a CPU-only box and exports every symbol include/*.h declares.
No compute calls here (no GPU)."""
import ctypes
import glob
import os
import re

from tests.conftest import ROOT


def declared_symbols():
    """extern "C" entry points declared in the C-ABI header (mixdq_math.h holds inline functions of
    the arithmetic specification, not exports)."""
    names = []
    for h in glob.glob(os.path.join(ROOT, "include", "mixdq_hip.h")):
        text = re.sub(r"/\*.*?\*/", "", open(h).read(), flags=re.S)
        names += re.findall(r"\b(mixdq_[a-z0-9_]+)\s*\(", text)
    return sorted(set(names))


def test_header_declares_the_operator_entry_points():
    syms = declared_symbols()
    for required in ("mixdq_quantize_f16_i8", "mixdq_qlinear_w8a8", "mixdq_qlinear_w8a8_rows",
                     "mixdq_qconv2d_w8a8", "mixdq_qconv2d_w8a8_table", "mixdq_conv_border_table",
                     "mixdq_conv_zero_point_propagate", "mixdq_qconv2d_workspace_bytes",
                     "mixdq_gemm_f16", "mixdq_status_string", "mixdq_abi_version",
                     "mixdq_igemm_select", "mixdq_groupnorm_silu_quantize",
                     "mixdq_groupnorm_workspace_bytes", "mixdq_layernorm_quantize",
                     "mixdq_geglu_quantize"):
        assert required in syms


def test_library_loads_and_exports_every_declared_symbol():
    from mixdq_amd.build import build
    lib = ctypes.CDLL(build())
    for name in declared_symbols():
        assert hasattr(lib, name), f"{name} declared in include/ but not exported"
    lib.mixdq_abi_version.restype = ctypes.c_int
    assert lib.mixdq_abi_version() == 1
    lib.mixdq_status_string.restype = ctypes.c_char_p
    assert lib.mixdq_status_string(0) == b"ok"
    assert b"alignment not to 4" in lib.mixdq_status_string(2)
    # one message per cause of "unsupported" (dilation is the only one the reference has)
    msgs = [lib.mixdq_status_string(c) for c in (3, 5, 6, 7, 8, 9)]
    assert b"dilation must be 1" in msgs[0] and len(set(msgs)) == 6
    assert b"K % 32" in msgs[1] and b"N % 64" in msgs[2] and b"padding" in msgs[3]
    lib.mixdq_qconv2d_workspace_bytes.restype = ctypes.c_size_t
    assert lib.mixdq_qconv2d_workspace_bytes(1280, 3, 3, 1) == 81 * 1280 * 4
    assert lib.mixdq_qconv2d_workspace_bytes(1280, 1, 1, 0) == 0


def test_argument_validation_needs_no_gpu():
    """Null pointers / bad sizes are rejected on the host before any launch."""
    from mixdq_amd.build import build
    lib = ctypes.CDLL(build())
    vp, i64, i32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int
    lib.mixdq_qlinear_w8a8.argtypes = [vp, vp, vp, vp, vp, vp, i64, i32, i32, i32, vp]
    assert lib.mixdq_qlinear_w8a8(None, None, None, None, None, None, 4, 8, 16, 0, None) == 1
    assert lib.mixdq_qlinear_w8a8(None, None, None, None, None, None, 0, 8, 16, 0, None) == 0
    assert lib.mixdq_qlinear_w8a8(None, None, None, None, None, None, -1, 8, 16, 0, None) == 1
    lib.mixdq_quantize_f16_i8.argtypes = [vp, vp, vp, vp, vp, i32, vp, vp, i32, vp]
    assert lib.mixdq_quantize_f16_i8(None, None, None, None, None, 9, None, None, 0, None) == 1


def test_product_path_has_no_oracle_or_cpu_fallback():
    """Nothing under mixdq_amd/ may import the oracle (it is test infrastructure)."""
    for path in glob.glob(os.path.join(ROOT, "mixdq_amd", "**", "*.py"), recursive=True):
        src = open(path).read()
        assert "import oracle" not in src and "from oracle" not in src, path
