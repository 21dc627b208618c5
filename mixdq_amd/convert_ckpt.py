"""qdiff `ckpt.pth` -> kernel `new_ckpt.pth`: counterpart of kernels/convert_ckpt.py:17-46.

    python -m mixdq_amd.convert_ckpt --ckpt ckpt.pth --save_path ./output

Input  (scripts/ptq.py:237-238, quant_model.py:126-131):
    {quantizer_name: [buffers OrderedDict(delta_list, zero_point_list, delta, zero_point, alpha),
                      params OrderedDict()]}
Output: {quantizer_name: {"delta_list": f16, "zero_point_list": f16}}
    weight quantizers reshaped to [3, OC], activation quantizers to [3]; the attention
    act_quantizer_q / _k / _v entries are dropped (never used by the kernels, convert_ckpt.py:23).
"""
from __future__ import annotations

import argparse
import os
from collections import OrderedDict

import torch

DROPPED = ("act_quantizer_k", "act_quantizer_q", "act_quantizer_v")
KEPT = ("delta_list", "zero_point_list")


def convert(checkpoint) -> "OrderedDict[str, dict]":
    out = OrderedDict()
    for key, value in checkpoint.items():
        if any(d in key for d in DROPPED):
            continue
        entry = {}
        buffers = value[0] if isinstance(value, (list, tuple)) else value
        if isinstance(buffers, dict):
            for sub in KEPT:
                t = buffers.get(sub)
                if not isinstance(t, torch.Tensor):
                    continue
                h = t.half()
                if "weight_quantizer" in key:
                    h = h.reshape(t.shape[0], t.shape[1])
                if "act_quantizer" in key:
                    h = h.reshape(t.shape[0])
                entry[sub] = h
        out[key] = entry
    return out


def main():
    ap = argparse.ArgumentParser(description="Convert a qdiff PTQ checkpoint for the INT8 kernels")
    ap.add_argument("--ckpt", type=str, required=True)
    ap.add_argument("--save_path", type=str, default="./output")
    args = ap.parse_args()
    new = convert(torch.load(args.ckpt, map_location="cpu"))
    os.makedirs(args.save_path, exist_ok=True)
    path = os.path.join(args.save_path, "new_ckpt.pth")
    torch.save(new, path)
    print(len(new), "quantizers ->", path)


if __name__ == "__main__":
    main()
