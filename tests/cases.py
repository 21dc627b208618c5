"""Shared operator-level test cases: shapes, seeds and deterministic input builders.

Used by tests/golden/gen_golden.py (build container, writes the fixtures) and by the CPU/GPU
tests (which regenerate the same inputs from the same seeds).  The first qlinear cases and the
first 15 conv cases are the shapes of the reference's own self-tests (op/qlinear.py:108,
op/qconv2d.py:104-119).
"""
import numpy as np
import torch
import torch.nn as nn

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




# ---- module-level cases (run through the reference's own classes in gen_golden.py) ----------
def make_float_module(c):
    kw, kind, name, seed = c, c["kind"], c["name"], c["seed"]
    if kind == "linear":
        m = nn.Linear(kw["cin"], kw["cout"], bias=kw["bias"])
        wshape = (kw["cout"], kw["cin"])
    else:
        m = nn.Conv2d(kw["cin"], kw["cout"], kw["ksize"], kw["stride"], kw["pad"], bias=kw["bias"])
        wshape = (kw["cout"], kw["cin"], kw["ksize"], kw["ksize"])
    w = dd.normal_f16(seed, wshape, std=0.05)
    m.weight.data = torch.from_numpy(w.astype(np.float32))
    if kw["bias"]:
        m.bias.data = torch.from_numpy(dd.normal_f16(seed + 1, (kw["cout"],), std=0.1).astype(np.float32))
    m.module_name = name
    return m


MODULE_CASES = [
    dict(key="lin_basic", kind="linear", name="down_blocks.1.attentions.0.transformer_blocks.0.attn1.to_q",
         cin=64, cout=32, bias=True, xshape=(2, 5, 64), seed=401),
    dict(key="lin_nobias", kind="linear", name="mid_block.attentions.0.transformer_blocks.0.attn1.to_k",
         cin=128, cout=48, bias=False, xshape=(1, 9, 128), seed=402),
    dict(key="lin_bos", kind="linear", name="down_blocks.1.attentions.0.transformer_blocks.0.attn2.to_k",
         cin=2048, cout=640, bias=False, xshape=(1, 77, 2048), seed=403, bos=True),
    dict(key="conv_p1", kind="conv", name="down_blocks.0.resnets.0.conv1",
         cin=64, cout=96, ksize=3, stride=1, pad=1, bias=True, xshape=(2, 64, 8, 8), seed=404),
    dict(key="conv_s2", kind="conv", name="down_blocks.0.downsamplers.0.conv",
         cin=64, cout=64, ksize=3, stride=2, pad=1, bias=True, xshape=(1, 64, 8, 8), seed=405),
    dict(key="conv_1x1", kind="conv", name="down_blocks.1.resnets.0.conv_shortcut",
         cin=64, cout=96, ksize=1, stride=1, pad=0, bias=True, xshape=(2, 64, 6, 6), seed=406),
    dict(key="conv_split", kind="conv", name="up_blocks.0.resnets.0.conv_shortcut",
         cin=96, cout=32, ksize=1, stride=1, pad=0, bias=True, xshape=(1, 96, 6, 6), seed=407,
         split=64),
]




def module_input(c):
    return torch.from_numpy(dd.normal_f16(c["seed"] + 10, c["xshape"], std=1.2))


def module_ckpt(c, golden):
    """Rebuild the kernel-format checkpoint of one module case from modules.npz."""
    key, name = c["key"], c["name"]
    ck = {}
    for sfx in ("weight_quantizer", "act_quantizer", "weight_quantizer_0", "act_quantizer_0"):
        k = f"{key}.ckpt.{sfx}.delta_list"
        if k in golden:
            ck[f"{name}.{sfx}"] = {
                "delta_list": torch.from_numpy(golden[k]),
                "zero_point_list": torch.from_numpy(golden[f"{key}.ckpt.{sfx}.zero_point_list"]),
            }
    return ck
