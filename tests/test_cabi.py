"""The C-ABI library loads on a CPU-only box and exports every symbol include/*.h declares.
No compute calls here (no GPU)."""
import ctypes
import glob
import os
import re

import pytest

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
    assert lib.mixdq_abi_version() == 3
    lib.mixdq_status_string.restype = ctypes.c_char_p
    assert lib.mixdq_status_string(0) == b"ok"
    assert b"alignment not to 4" in lib.mixdq_status_string(2)
    # one message per cause of "unsupported" (dilation is the only one the reference has)
    msgs = [lib.mixdq_status_string(c) for c in (3, 5, 6, 7, 8, 9)]
    assert b"dilation must be 1" in msgs[0] and len(set(msgs)) == 6
    assert b"K % 32" in msgs[1] and b"N % 32" in msgs[2] and b"padding" in msgs[3]
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


def test_tile_choice_is_a_host_function_of_the_shape():
    """The automatic tile choice (host code, no GPU): UNet shapes at batch 1 and batch 8, the
    Linear-only four-phase loop, the GEMM+GEGLU launch and the packed-W4 rule."""
    from mixdq_amd.build import build
    lib = ctypes.CDLL(build())
    i64, i32 = ctypes.c_int64, ctypes.c_int
    for fn in (lib.mixdq_igemm_select_id, lib.mixdq_igemm_select_id_w4):
        fn.argtypes, fn.restype = [i64, i32, i32, i32], i32
    lib.mixdq_igemm_select_id_geglu.argtypes, lib.mixdq_igemm_select_id_geglu.restype = [i64, i32, i32, i32], i32
    sel = lib.mixdq_igemm_select_id
    # batch 1: exactly one workgroup per CU (64x80), six stages for cold K <= 2048, four beyond
    assert sel(1024, 1280, 1280, 1280) == 56 and sel(1024, 1280, 5120, 5120) == 45
    assert sel(1024, 10240, 1280, 1280) == 27 and sel(4096, 640, 2560, 2560) == 44
    assert sel(8192, 1280, 5120, 5120) == 28 and sel(8192, 1280, 1280, 1280) == 27      # 128x320: 4 x 4 waves for long K
    # batch 8: the four-phase 256x256 loop for plain Linear launches from 1.5 workgroups per CU on ...
    # (71 = that tile's persistent form, one workgroup per CU walking its tiles -- csrc/igemm_pp.h: what the rule
    #  takes wherever a CU has more than one 256x256 tile; MIXDQ_IGEMM_PERSIST=0 gives 70 back)
    assert sel(8192, 10240, 1280, 1280) == 71 and sel(8192, 3840, 1280, 1280) == 71
    assert sel(32768, 1920, 640, 640) == 71
    # ... not for convolutions (k_align = C != k_total: the gather needs the general staging) ...
    assert sel(32768, 640, 640, 5760) not in (70, 71) and sel(8192, 10240, 1280, 11520) not in (70, 71)
    # ... nor a K that is not whole 128-byte tiles; GEMM + GEGLU does (its epilogue runs in registers)
    assert sel(8192, 10240, 1296, 1296) not in (70, 71)
    assert lib.mixdq_igemm_select_id_geglu(8192, 10240, 1280, 0) == 71
    assert lib.mixdq_igemm_select_id_geglu(1024, 10240, 1280, 0) == 27
    assert lib.mixdq_igemm_select_id_geglu(1024, 10240 + 16, 1280, 0) == -1      # N % 32 != 0
    # packed W4: 32x32x32 tiles (every wave unpacks what it multiplies)
    assert lib.mixdq_igemm_select_id_w4(1024, 1280, 1280, 1280) == 41
    assert lib.mixdq_igemm_select_id_w4(1024, 1280, 1281, 1281) == -1             # K % 32 != 0
    assert sel(64, 8, 20, 20) == 0 and sel(64, 6, 16, 16) == -1                   # generic / invalid


def test_aq_kernels_never_touch_a_register_whose_load_is_in_flight():
    """csrc/igemm_aq.hip requests its FP16 operand with inline-asm loads that the compiler cannot see in flight
    (its own wait-count pass serialised the prefetch pipeline); tools/check_aq_isa.py proves on the generated ISA
    of every AQ kernel that between a request and its counted wait nothing else mentions those registers (no
    copy, no spill), that prologue and loop use the same physical registers, and that the loop tail is drained."""
    import shutil
    import subprocess
    import sys
    hipcc = "/opt/rocm/bin/hipcc" if os.path.exists("/opt/rocm/bin/hipcc") else shutil.which("hipcc")
    if not hipcc:
        pytest.skip("no hipcc on this box (the check runs where the library is built)")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_aq_isa.py")], capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-1000:]
    assert "20 AQ kernels checked, 0 failed" in r.stdout


def test_f16in_auto_follows_the_measured_records():
    """mixdq_qlinear_f16in_preferred decides the DEFAULT launch form of every drop-in Linear / 1x1 conv.  It is a
    table of measurements (csrc/f16in_table.h <- tools/gen_f16in_table.py <- the committed records of
    tools/bench_f16in.py), not a fitted model: on EVERY recorded row the form it picks is not slower
    than the other by more than 3 % in any record of that shape; the generated header is what the generator makes
    from the records today; unmeasured (N, K) pairs keep the reference's two launches."""
    import json
    import subprocess
    import sys
    from mixdq_amd.build import build
    lib = ctypes.CDLL(build())
    pref = lib.mixdq_qlinear_f16in_preferred
    pref.argtypes, pref.restype = [ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int], ctypes.c_int
    records = {}
    for rel in ("profiles/r05_f16in_per_layer.txt", "profiles/r06_f16in_per_layer.txt"):
        path = os.path.join(ROOT, rel)
        if not os.path.exists(path):
            continue
        for ln in open(path):
            if ln.startswith("{"):
                r = json.loads(ln)
                records.setdefault((r["M"], r["N"], r["K"]), []).append((r["pair_us"], r["f16in_us"]))
    assert len(records) >= 18
    for (M, N, K), recs in records.items():
        if pref(M, N, K, 0):     # the one launch is only taken where it was not slower (3 %) in ANY record of the shape
            assert all(f <= 1.03 * p_ for p_, f in recs), (M, N, K, recs)
        else:                    # the reference's two launches: kept unless the one launch won EVERY record
            assert any(p_ <= f for p_, f in recs), (M, N, K, recs)
    assert pref(1024, 1280, 640, 0) == 1 and pref(4096, 5120, 640, 0) == 0        # the clearest rows of either sign
    assert pref(1024, 1296, 640, 0) == 0 and pref(1024, 1280, 656, 0) == 0         # never measured: the reference's flow
    assert pref(16384 * 3, 320, 640, 0) == 1 and pref(16384 * 40, 320, 640, 0) == 0    # M within / beyond a factor of two of a row
    hdr = os.path.join(ROOT, "mixdq_amd", "csrc", "f16in_table.h")
    before = open(hdr).read()
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "gen_f16in_table.py")], stdout=subprocess.DEVNULL)
    assert open(hdr).read() == before, "csrc/f16in_table.h is stale: run tools/gen_f16in_table.py"


def test_persistent_form_is_a_switchable_choice_of_the_rule():
    """MIXDQ_IGEMM_PERSIST=0 (read once per process) gives the one-workgroup-per-tile four-phase kernel (70) back where
    the rule otherwise takes its persistent form (71); launches with at most one 256x256 tile per CU never take 71."""
    import subprocess
    import sys
    code = ("import ctypes; from mixdq_amd.build import build; lib = ctypes.CDLL(build()); "
            "f = lib.mixdq_igemm_select_id; f.argtypes = [ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int]; "
            "print('IDS', f(8192, 10240, 1280, 1280), f(8192, 3840, 1280, 1280), f(2048, 10240, 1280, 1280))")
    ids = {}
    for flag in ("1", "0"):
        r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=dict(os.environ, MIXDQ_IGEMM_PERSIST=flag),
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-2000:]
        ids[flag] = [int(v) for v in [ln for ln in r.stdout.splitlines() if ln.startswith("IDS ")][-1].split()[1:]]
    assert ids["1"][:2] == [71, 71] and ids["0"][:2] == [70, 70]
    assert ids["1"][2] == ids["0"][2] and ids["1"][2] not in (70, 71)       # (2048, 10240, 1280): 320 tiles of 256x256 < 1.5 per CU
