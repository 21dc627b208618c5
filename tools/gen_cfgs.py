#!/usr/bin/env python3
"""Pack the reference's per-layer bit-width maps (kernels/cfgs/{weight,act}/*.yaml -- data files,
the outputs of its offline mixed-precision search) into mixdq_amd/cfgs/bitwidths.json.

Runs only in the build container (needs /root/reference).  One string per config, one character
per layer in mixdq_amd.unet inventory order: '2' / '4' / '8' = bits, '-' = layer absent from that
yaml (activation stays FP16).  The BOS rows (kernels/bos_pre_computed.pt) are weights-derived
tensors of the real SDXL-Turbo checkpoint; only their shapes are recorded (no checkpoint here).
"""
import glob
import hashlib
import json
import os
import sys

import torch
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = os.environ.get("MIXDQ_REFERENCE", "/root/reference")

from mixdq_amd.unet import SDXLUNet, quantizable_layers  # noqa: E402


def main():
    with torch.device("meta"):
        names = list(quantizable_layers(SDXLUNet()))
    out = dict(n_layers=len(names),
               names_sha256=hashlib.sha256("\n".join(names).encode()).hexdigest(), configs={})
    for path in sorted(glob.glob(os.path.join(REF, "kernels", "cfgs", "*", "*.yaml"))):
        key = "/".join(path.split(os.sep)[-2:])[:-5]
        d = {(k[6:] if k.startswith("model.") else k): v for k, v in yaml.safe_load(open(path)).items()}
        assert set(d) <= set(names), key
        out["configs"][key] = "".join(str(d[n]) if n in d else "-" for n in names)
    bos = torch.load(os.path.join(REF, "kernels", "bos_pre_computed.pt"), map_location="cpu")
    out["bos_shapes"] = {k: list(v.shape) for k, v in bos.items()}
    dst = os.path.join(ROOT, "mixdq_amd", "cfgs", "bitwidths.json")
    with open(dst, "w") as f:
        json.dump(out, f, indent=0)
    print(dst, os.path.getsize(dst), "bytes;", len(out["configs"]), "configs")


if __name__ == "__main__":
    main()
