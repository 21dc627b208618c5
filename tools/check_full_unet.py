#!/usr/bin/env python3
"""Full-size wiring check (SDXL-Turbo UNet, 1024 px): the fused W8A8 graph (producer fusions, packed
q|k|v, grouped k|v / time-embedding launches, GEMM+GEGLU, fused to_q + cross-attention, two-source
GroupNorm, FP16 layers on own kernels, hipGraph) against the unfused drop-in graph and the FP16
network, batch 1 and 2.  Prints the distances; a wiring mistake is an error of the order of the
output's spread, the fusions differ at quantization-noise level.

    python tools/check_full_unet.py
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from bench import Cfg  # noqa: E402
from mixdq_amd import cfgs  # noqa: E402
from mixdq_amd.calib import calibrate, precompute_bos  # noqa: E402
from mixdq_amd.quantize_sdxl import example_inputs, hip_graph_opt, quantize_unet  # noqa: E402
from mixdq_amd.unet import build_unet  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    ok = True
    for B in (1, 2):
        unet = build_unet(dev)
        inputs = example_inputs(B, 128, dev, seed=42)
        with torch.no_grad():
            ref = unet(**inputs)[0].float()
        ckpt = calibrate(unet, [inputs])
        bos = precompute_bos(unet, inputs["encoder_hidden_states"])
        quantize_unet(unet, Cfg(cfgs.load("weight/uniform_8"), cfgs.load("act/act_8.00")), ckpt,
                      bos=True, bos_dict=bos)
        with torch.no_grad():
            unfused = unet(**inputs)[0].float()
            unet.set_fused(True)
            fused = unet(**inputs)[0].float()
            again = unet(**inputs)[0].float()
            hip_graph_opt(unet)
            graphed = unet(**inputs)[0].float()
        spread = ref.std().item()
        n_mean = (unfused - ref).abs().mean().item()
        f_mean = (fused - ref).abs().mean().item()
        d_mean = (fused - unfused).abs().mean().item()
        d_max = (fused - unfused).abs().max().item()
        det = bool(torch.equal(fused, again) and torch.equal(fused, graphed))
        good = det and f_mean <= 1.25 * n_mean + 1e-3 and d_mean <= n_mean and torch.isfinite(fused).all()
        ok &= bool(good)
        print(f"batch {B}: output spread (std) {spread:.4f} | unfused-vs-FP16 mean {n_mean:.5f} | "
              f"fused-vs-FP16 mean {f_mean:.5f} | fused-vs-unfused mean {d_mean:.5f} max {d_max:.4f} | "
              f"deterministic + graph replay identical: {det} | {'OK' if good else 'FAIL'}")
        del unet
        torch.cuda.empty_cache()
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
