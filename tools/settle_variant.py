#!/usr/bin/env python3
"""Settle the one point this build cannot observe: did nvcc contract  x * s_inv + zp  (quantizer)
and  (acc - bias0) * scale + bias  (epilogue) of the reference's CUDA kernels into an FMA
("variant A", this build's default) or not ("variant B", MIXDQ_EPILOGUE_VARIANT=B)?

For a maintainer with an NVIDIA GPU and the reference's built extension:

    cd <MixDQ checkout>/kernels && pip install -e .      # builds mixdq_extension._C
    python <this repo>/tools/settle_variant.py

It feeds the reference's own operators (kernels/mixdq_extension/op/quant.py:7-30,
op/qlinear.py:28-108: `mixdq_extension._C.quantize_per_tensor_to_int8`, `qlinear_w8_a8_ohalf`) the
inputs on which the two variants differ -- the 36 (x, scale_inv, zero_point) separator triples of
tests/golden/ops_small.npz (`q_sep_*`) and the qlinear cases of tests/golden/ops.json with
`n_diff_A_vs_B > 0` -- and prints which variant the binary implements.  Pure NumPy + the committed
fixtures (+ torch and the reference extension on the maintainer's box); it uses nothing of this
repo's kernels or oracle and never runs on the MI355X box.
"""
import argparse
import hashlib
import importlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import detdata as dd  # noqa: E402  (pure NumPy input generator of the fixtures)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def linear_inputs(case):
    """Inputs of a full-range qlinear case, as tests/cases.py builds them (seeds in ops.json)."""
    M, K, N, seed = case["M"], case["K"], case["N"], case["seed"]
    lo, hi = case["wrange"]
    assert (lo, hi) == (-128, 128), "separating cases are the full-range ones"
    w = dd.int8(seed, (N, K), lo, hi)
    a = dd.int8(seed + 1000, (M, K))
    wscale = (dd.f32(seed + 2000, (N,)) + np.float32(0.1)).astype(np.float32) * np.float32(0.01)
    in_scale, in_zp = np.float32(0.0312), np.float32(-11.0)
    bias = dd.f16(seed + 3000, (N,)) if case["bias"] else None
    wsum = w.astype(np.float32).sum(axis=1, dtype=np.float32)
    scale = (wscale * in_scale).astype(np.float32)
    bias0 = (wsum * in_zp).astype(np.float32)
    return a, w, wscale, in_scale, in_zp, wsum, scale, bias0, bias


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawTextHelpFormatter)
    ap.add_argument("--ext", default="mixdq_extension._C",
                    help="module that exports the reference's operators")
    ap.add_argument("--device", default="cuda")
    args = ap.parse_args()
    try:
        import torch
        ext = importlib.import_module(args.ext)
    except ImportError as e:
        sys.exit(f"cannot import torch / {args.ext}: {e}\nbuild the reference's extension first "
                 "(kernels/setup.py) and run this on the box that has it")
    dev = args.device
    small = dict(np.load(os.path.join(GOLDEN, "ops_small.npz")))
    with open(os.path.join(GOLDEN, "ops.json")) as f:
        ops = json.load(f)
    verdicts = []

    # ---- quantizer: one triple per call (scale_inv / zero_point are 0-dim device tensors) -------
    votes = {"A": 0, "B": 0, "neither": 0}
    for x, si, zp, qa, qb in zip(small["q_sep_x"], small["q_sep_sinv"], small["q_sep_zp"],
                                 small["q_sep_A"], small["q_sep_B"]):
        xin = torch.full((64,), float(x), dtype=torch.float16, device=dev)
        for fn in (ext.quantize_per_tensor_to_int8, ext.quantize_per_tensor_to_int8_vectorized):
            got = fn(xin, torch.tensor(float(si), device=dev), torch.tensor(float(zp), device=dev))
            got = set(got.cpu().numpy().astype(np.int8).tolist())
            assert len(got) == 1, "the operator is not elementwise?"
            g = got.pop()
            votes["A" if g == qa else "B" if g == qb else "neither"] += 1
    print(f"quantize (36 separator triples x 2 entry points): {votes}")
    verdicts.append(max(votes, key=votes.get) if votes["neither"] == 0 and
                    min(votes["A"], votes["B"]) == 0 else "mixed")

    # ---- qlinear epilogue: whole cases, by SHA-256 of the fp16 output -------------------------
    for case in ops["qlinear"]:
        if case["n_diff_A_vs_B"] == 0:
            continue
        a, w, wscale, in_scale, in_zp, wsum, scale, bias0, bias = linear_inputs(case)
        t = lambda v: None if v is None else torch.from_numpy(np.ascontiguousarray(v)).to(dev)  # noqa: E731
        out = ext.qlinear_w8_a8_ohalf(t(a), t(w), t(wscale), torch.tensor(float(in_scale), device=dev),
                                      torch.tensor(float(in_zp), device=dev), t(wsum), t(scale),
                                      t(bias0), t(bias))
        h = sha(out.cpu().numpy())
        which = "A" if h == case["sha_A"] else "B" if h == case["sha_B"] else "neither"
        print(f"qlinear {case['name']} ({case['n_diff_A_vs_B']} separating outputs): {which}")
        verdicts.append(which)

    if all(v == "A" for v in verdicts):
        print("VERDICT: variant A (FMA) -- this build's default matches the CUDA binary")
    elif all(v == "B" for v in verdicts):
        print("VERDICT: variant B (mul, then add) -- run this build with MIXDQ_EPILOGUE_VARIANT=B")
    else:
        print(f"VERDICT: inconclusive {verdicts}: quantizer and epilogue were compiled differently, "
              "or an output matches neither variant (please report the lines above)")


if __name__ == "__main__":
    main()
