#!/usr/bin/env python3
"""Root cause of round 1's red GPU test (tests/test_modules_gpu.py::
test_fused_path_with_fp16_fallback_layers, 0.0543 vs a hand-picked bound of 0.0536 on the driver's
box, green on the builder's): N repetitions of the OLD procedure (calibration on the GPU FP16
network, error of the fused graph vs the FP16 network against 0.05 * max + 0.02) next to the NEW
one (calibration on a CPU FP32 copy, fused vs unfused graph with the same fallbacks, bound relative
to the measured quantization noise).  Prints one line per repetition and the spread.

    python tools/flake_probe.py [N]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from tests.test_host import TINY, Args, tiny_inputs  # noqa: E402

DEV = "cuda:0"
DROP = {"conv_in", "conv_out", "down_blocks.0.resnets.0.conv2",
        "down_blocks.1.attentions.0.transformer_blocks.0.ff.net.2",
        "down_blocks.1.attentions.0.transformer_blocks.0.attn1.to_k",
        "down_blocks.1.attentions.0.proj_in"}


def to_dev(h):
    return dict(sample=h["sample"].half().to(DEV), timestep=h["timestep"].to(DEV),
                encoder_hidden_states=h["encoder_hidden_states"].half().to(DEV),
                added_cond_kwargs={k: v.half().to(DEV) for k, v in h["added_cond_kwargs"].items()})


def one(calib_on_gpu: bool):
    from mixdq_amd.calib import calibrate, precompute_bos
    from mixdq_amd.quantize_sdxl import quantize_unet
    from mixdq_amd.unet import build_unet, quantizable_layers
    host = tiny_inputs(B=1, L=16)
    unet = build_unet(DEV, cfg=TINY)
    inp = to_dev(host)
    with torch.no_grad():
        ref = unet(**inp)[0].float()
        if calib_on_gpu:
            ckpt = calibrate(unet, [inp])
            bos = precompute_bos(unet, inp["encoder_hidden_states"])
        else:
            cpu = build_unet("cpu", dtype=torch.float32, cfg=TINY)
            ckpt = calibrate(cpu, [host])
            bos = {k: v.half().to(DEV) for k, v in
                   precompute_bos(cpu, host["encoder_hidden_states"]).items()}
    names = list(quantizable_layers(unet))
    quantize_unet(unet, Args({"model." + n: 8 for n in names},
                             {"model." + n: 8 for n in names if n not in DROP}),
                  ckpt, bos=True, bos_dict=bos)
    with torch.no_grad():
        unfused = unet(**inp)[0].float()
        unet.set_fused(True)
        fused = unet(**inp)[0].float()
    return dict(err_vs_fp16=(fused - ref).abs().max().item(),
                old_bound=0.05 * ref.abs().max().item() + 0.02,
                noise_max=(unfused - ref).abs().max().item(),
                noise_mean=(unfused - ref).abs().mean().item(),
                d_max=(fused - unfused).abs().max().item(),
                d_mean=(fused - unfused).abs().mean().item())


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    for label, gpu_cal in (("old: GPU-FP16 calibration", True), ("new: CPU-FP32 calibration", False)):
        rows = [one(gpu_cal) for _ in range(n)]
        print(f"--- {label}, {n} repetitions")
        for r in rows:
            print("  err_vs_fp16 %.5f (old bound %.5f, margin %+.5f) | fused-unfused max %.5f mean %.6f"
                  " | noise max %.5f mean %.6f" % (r["err_vs_fp16"], r["old_bound"],
                                                   r["old_bound"] - r["err_vs_fp16"], r["d_max"],
                                                   r["d_mean"], r["noise_max"], r["noise_mean"]))
        errs = [r["err_vs_fp16"] for r in rows]
        print("  err_vs_fp16 min %.5f max %.5f distinct %d; old-bound failures %d / %d" % (
            min(errs), max(errs), len(set(errs)), sum(r["err_vs_fp16"] >= r["old_bound"] for r in rows), n))
        print("  new criterion (d_max <= 1.5 noise_max, d_mean <= noise_mean) failures %d / %d" % (
            sum(not (r["d_max"] <= 1.5 * r["noise_max"] + 1e-3 and r["d_mean"] <= r["noise_mean"] + 1e-4)
                for r in rows), n))


if __name__ == "__main__":
    main()
