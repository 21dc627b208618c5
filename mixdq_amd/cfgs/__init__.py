"""Per-layer bit-width configurations (the reference's kernels/cfgs/{weight,act}/*.yaml, packed by
tools/gen_cfgs.py into bitwidths.json: one character per layer in UNet inventory order).

    load("weight/uniform_8")  -> {layer name: bits}
    load("act/act_8.00")      -> 785 layers (the 9 activation-protected ones are absent)
    write_yaml(name, path)     -> a yaml file in the reference's own schema ("model.<name>: bits")
"""
from __future__ import annotations

import hashlib
import json
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_cache = None


def _data():
    global _cache
    if _cache is None:
        with open(os.path.join(_HERE, "bitwidths.json")) as f:
            _cache = json.load(f)
    return _cache


def available():
    return sorted(_data()["configs"])


def layer_names():
    import torch
    from mixdq_amd.unet import SDXLUNet, quantizable_layers
    with torch.device("meta"):
        names = list(quantizable_layers(SDXLUNet()))
    d = _data()
    assert len(names) == d["n_layers"]
    assert hashlib.sha256("\n".join(names).encode()).hexdigest() == d["names_sha256"], \
        "UNet layer inventory no longer matches the packed bit-width configs"
    return names


def load(name: str) -> dict:
    packed = _data()["configs"][name]
    return {n: int(c) for n, c in zip(layer_names(), packed) if c != "-"}


def write_yaml(name: str, path: str, prefix: str = "model.") -> str:
    with open(path, "w") as f:
        for layer, bits in load(name).items():
            f.write(f"{prefix}{layer}: {bits}\n")
    return path


def bos_shapes() -> dict:
    return {k: tuple(v) for k, v in _data()["bos_shapes"].items()}
