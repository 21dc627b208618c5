"""Which PyTorch glue kernels (copies, cats, adds ...) still run in one fused W8A8 UNet forward, by
input shape: the to-do list for further producer/epilogue fusions.  GPU only."""
import argparse
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--px", type=int, default=1024)
    ap.add_argument("--batch", type=int, default=1)
    a = ap.parse_args()
    from bench import Cfg
    from mixdq_amd import cfgs
    from mixdq_amd.calib import calibrate, precompute_bos
    from mixdq_amd.quantize_sdxl import example_inputs, quantize_unet
    from mixdq_amd.unet import build_unet
    dev = torch.device("cuda:0")
    unet = build_unet(dev)
    inputs = example_inputs(a.batch, a.px // 8, dev, seed=42)
    ckpt = calibrate(unet, [inputs], bos=True)
    bos = precompute_bos(unet, inputs["encoder_hidden_states"])
    quantize_unet(unet, Cfg(cfgs.load("weight/uniform_8"), cfgs.load("act/act_8.00")), ckpt, bos=True,
                  bos_dict=bos)
    unet.set_fused(True)
    for _ in range(2):
        unet(**inputs)
    torch.cuda.synchronize()
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True,
                 with_stack=True) as prof:
        unet(**inputs)
        torch.cuda.synchronize()
    agg = collections.defaultdict(lambda: [0, 0.0, None])
    for e in prof.events():
        if e.name in ("aten::copy_", "aten::cat", "aten::add", "aten::add_", "aten::silu",
                      "aten::mul", "aten::upsample_nearest2d", "aten::_to_copy", "aten::clone"):
            key = (e.name, str(e.input_shapes)[:90])
            stack = [s for s in (e.stack or []) if "mixdq_amd" in s or "bench" in s]
            agg[key][0] += 1
            agg[key][1] += e.device_time_total
            agg[key][2] = stack[0] if stack else None
    rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
    for (name, shapes), (n, us, where) in rows[:40]:
        print(f"{us:9.1f} us {n:4d}x {name:24s} {shapes}  <- {where}")


if __name__ == "__main__":
    main()
