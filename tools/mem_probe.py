#!/usr/bin/env python3
"""Where the static memory of the quantized network goes: bytes held by module buffers (by kind) vs
torch.cuda.memory_allocated()."""
import collections, gc, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import Cfg
from mixdq_amd import cfgs
from mixdq_amd.calib import calibrate, precompute_bos
from mixdq_amd.quantize_sdxl import example_inputs, quantize_unet
from mixdq_amd.unet import build_unet

dev = torch.device("cuda", 0)
unet = build_unet(dev)
print("fp16 resident MB", torch.cuda.memory_allocated() / 2**20)
inputs = example_inputs(1, 128, dev, seed=42)
ckpt = calibrate(unet, [inputs])
bos = precompute_bos(unet, inputs["encoder_hidden_states"])
quantize_unet(unet, Cfg(cfgs.load("weight/uniform_8"), cfgs.load("act/act_8.00")), ckpt, bos=True, bos_dict=bos)
del ckpt
gc.collect(); torch.cuda.empty_cache()
print("after quantize MB", torch.cuda.memory_allocated() / 2**20)
unet.set_fused(True)
with torch.no_grad():
    unet(**inputs)
torch.cuda.synchronize(); gc.collect(); torch.cuda.empty_cache()
print("after fused forward MB", torch.cuda.memory_allocated() / 2**20)
by = collections.Counter()
seen = set()
for name, b in list(unet.named_buffers()) + list(unet.named_parameters()):
    st = b.untyped_storage()
    if st.data_ptr() in seen:
        continue
    seen.add(st.data_ptr())
    by[name.rsplit(".", 1)[-1] + ":" + str(b.dtype)] += st.nbytes()
for k, v in by.most_common(12):
    print(f"  {k:50s} {v / 2**20:9.1f} MB")
print("sum of distinct storages MB", sum(by.values()) / 2**20)
extra = 0
for m in unet.modules():
    for key in ("_tables", "_kv_buf", "_kvpack", "_qkv", "_w_padded"):
        v = m.__dict__.get(key)
        if v is None:
            continue
        stack = [v]
        while stack:
            o = stack.pop()
            if torch.is_tensor(o):
                if o.untyped_storage().data_ptr() not in seen:
                    seen.add(o.untyped_storage().data_ptr()); extra += o.untyped_storage().nbytes()
            elif isinstance(o, dict):
                stack += list(o.values())
            elif isinstance(o, (tuple, list)):
                stack += list(o)
print("derived caches MB", extra / 2**20)
