"""Shared operator-level test cases: shapes, seeds and deterministic input builders.

Used by tests/golden/gen_golden.py (build container, writes the fixtures) and by the CPU/GPU
tests (which regenerate the same inputs from the same seeds).  The first qlinear cases and the
first 15 conv cases are the shapes of the reference's own self-tests (op/qlinear.py:108,
op/qconv2d.py:104-119).
"""
import numpy as np

from oracle import oracle as O
from tests import detdata as dd

LINEAR_CASES = [
    # name, M, K, N, bias, (lo, hi) of weights, seed, note
    ("lin_ref_small_bias", 64, 8, 16, True, (-3, 3), 201, "op/qlinear.py:108 run_test(64, 8, 16)"),
    ("lin_ref_small_nobias", 64, 8, 16, False, (-3, 3), 202, "same, use_bias=False"),
    ("lin_attn2_to_k", 77, 2048, 640, False, (-128, 128), 203, "SDXL attn2.to_k (77,640,2048)"),
    ("lin_attn1_1280", 256, 1280, 1280, True, (-128, 128), 204, "SDXL attn1/proj (M scaled 1024->256)"),
    ("lin_ff_geglu", 128, 640, 5120, True, (-128, 128), 205, "SDXL ff.net.0.proj (T,8c,c), M scaled"),
    ("lin_ff_out", 96, 5120, 1280, True, (-128, 128), 206, "SDXL ff.net.2; |acc| > 2^24 exercises cvt RNE"),
    ("lin_time_emb", 1, 2816, 1280, True, (-128, 128), 207, "add_embedding.linear_1, M=1"),
    ("lin_ragged", 37, 48, 24, True, (-128, 128), 208, "K%16==0, N%8==0, ragged M"),
    ("lin_small_align", 19, 20, 12, True, (-128, 128), 209, "K%4, N%4 only: small-alignment path"),
    ("lin_sat", 8, 4096, 16, False, (127, 128), 210, "all weights 127, inputs -128: max |acc|"),
]


def linear_inputs(case):
    name, M, K, N, has_bias, (lo, hi), seed, _ = case
    w = dd.int8(seed, (N, K), lo, hi)
    if name == "lin_sat":
        a = np.full((M, K), -128, dtype=np.int8)
    elif lo == -3:
        x16 = dd.f16(seed + 1000, (M, K), -3.0, 3.0)
        a = O.quantize(x16, 0.123, 5.0)   # op/qlinear.py:46 passes input_scale (sic) as scale_inv
    else:
        a = dd.int8(seed + 1000, (M, K))
    wscale = (dd.f32(seed + 2000, (N,)) + np.float32(0.1)).astype(np.float32) * (
        np.float32(1.0) if lo == -3 else np.float32(0.01))
    in_scale = np.float32(0.123 if lo == -3 else 0.0312)
    in_zp = np.float32(5.0 if lo == -3 else -11.0)
    bias = dd.f16(seed + 3000, (N,)) if has_bias else None
    wsum = w.astype(np.float32).sum(axis=1, dtype=np.float32)
    scale = (wscale * in_scale).astype(np.float32)
    bias0 = (wsum * in_zp).astype(np.float32)
    return a, w, wscale, in_scale, in_zp, bias, scale, bias0



CONV_CASES = [
    # name, n,h,w,c,k,r,s,pad,stride,bias,(lo,hi),seed   -- first 15 = op/qconv2d.py:104-119
    ("conv_ref_00", 1, 14, 14, 512, 1024, 3, 3, 1, 1, True, (-3, 3), 301),
    ("conv_ref_01", 1, 14, 14, 512, 1024, 3, 3, 1, 1, True, (-3, 3), 302),
    ("conv_ref_02", 1, 14, 14, 512, 1024, 3, 3, 1, 2, True, (-3, 3), 303),
    ("conv_ref_03", 1, 14, 14, 512, 1024, 3, 3, 0, 1, False, (-3, 3), 304),
    ("conv_ref_04", 1, 14, 14, 512, 1024, 3, 3, 0, 1, True, (-3, 3), 305),
    ("conv_ref_05", 1, 14, 14, 512, 1024, 3, 3, 0, 2, True, (-3, 3), 306),
    ("conv_ref_06", 1, 14, 14, 512, 1024, 3, 3, 0, 1, False, (-3, 3), 307),
    ("conv_ref_07", 1, 7, 7, 4, 320, 3, 3, 1, 1, True, (-3, 3), 308),
    ("conv_ref_08", 1, 7, 7, 4, 320, 3, 3, 0, 1, True, (-3, 3), 309),
    ("conv_ref_09", 1, 7, 7, 4, 320, 3, 3, 1, 2, True, (-3, 3), 310),
    ("conv_ref_10", 1, 7, 7, 4, 320, 3, 3, 0, 2, True, (-3, 3), 311),
    ("conv_ref_11", 1, 7, 7, 320, 4, 3, 3, 1, 1, True, (-3, 3), 312),
    ("conv_ref_12", 1, 7, 7, 320, 4, 3, 3, 0, 1, True, (-3, 3), 313),
    ("conv_ref_13", 1, 7, 7, 320, 4, 3, 3, 1, 2, True, (-3, 3), 314),
    ("conv_ref_14", 1, 7, 7, 320, 4, 3, 3, 0, 2, True, (-3, 3), 315),
    # SDXL-shaped (scaled-down spatial), full-range int8
    ("conv_res_320", 2, 16, 16, 320, 320, 3, 3, 1, 1, True, (-128, 128), 320),
    ("conv_res_960_640", 1, 12, 12, 960, 640, 3, 3, 1, 1, True, (-128, 128), 321),
    ("conv_down_s2", 2, 16, 16, 320, 320, 3, 3, 1, 2, True, (-128, 128), 322),
    ("conv_shortcut_1x1", 2, 8, 8, 1920, 640, 1, 1, 0, 1, True, (-128, 128), 323),
    ("conv_odd_hw", 1, 9, 5, 64, 72, 3, 3, 1, 1, False, (-128, 128), 324),
    ("conv_tiny_hw", 3, 1, 2, 32, 16, 3, 3, 1, 1, True, (-128, 128), 325),
    ("conv_s2_odd", 1, 7, 9, 64, 40, 3, 3, 1, 2, True, (-128, 128), 326),
]


def conv_inputs(case):
    name, n, h, w, c, k, r, s, pad, stride, has_bias, (lo, hi), seed = case
    x = dd.int8(seed + 1000, (n, h, w, c), lo, hi)            # NHWC order
    wt = dd.int8(seed, (k, r, s, c), lo, hi)                   # KRSC order
    wscale = (dd.f32(seed + 2000, (k,)) + np.float32(0.1)).astype(np.float32) * (
        np.float32(1.0) if lo == -3 else np.float32(0.01))
    in_scale = np.float32(0.123 if lo == -3 else 0.0312)
    in_zp = np.float32(2.345 if lo == -3 else -11.0)           # op/qconv2d.py:43 uses 2.345
    bias = dd.f16(seed + 3000, (k,)) if has_bias else None
    scale = (wscale * in_scale).astype(np.float32)
    wsum = wt.astype(np.float32).sum(axis=3, dtype=np.float32)  # [K,R,S]
    bias0 = (wsum.reshape(k, -1).sum(axis=1, dtype=np.float32) * in_zp).astype(np.float32)
    return x, wt, wscale, in_scale, in_zp, bias, scale, wsum, bias0


