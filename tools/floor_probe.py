#!/usr/bin/env python3
"""What does a kernel boundary cost inside a captured graph on this box, and which HIP runtime knob
moves it?  Dependent chains of L launches in one hipGraph, microseconds per launch (HIP events):

  tiny       torch add_ on one element (the boundary itself)
  quant8     the quantize kernel on 8 elements (this library's smallest launch)
  ln         LayerNorm+quantize on [1024, 1280]
  gemm       the (1024, 1280, 1280) W8A8 Linear with 40 distinct weight tensors (cold, as in the UNet)
  pair       ln -> gemm (+ residual), the transformer chain's unit

    python tools/floor_probe.py            one line of JSON
    python tools/floor_probe.py --sweep    the same under a list of runtime environment knobs, each
                                           in a fresh child process (the knobs are read at HIP init)
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KNOBS = [
    {},
    {"HIP_FORCE_DEV_KERNARG": "1"},
    {"HIP_FORCE_DEV_KERNARG": "0"},
    {"AMD_OPT_FLUSH": "0"},
    {"AMD_OPT_FLUSH": "3"},
    {"DEBUG_CLR_GRAPH_PACKET_CAPTURE": "0"},
    {"DEBUG_HIP_GRAPH_BATCH_SIZE": "1024"},
    {"ROC_ACTIVE_WAIT_TIMEOUT": "1000"},
]


def sweep():
    for k in KNOBS:
        env = dict(os.environ, **k)
        r = subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, capture_output=True,
                           text=True, timeout=600)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        print(json.dumps(k), line[-1] if line else "FAILED " + r.stderr[-300:], flush=True)


def main():
    import torch
    sys.path.insert(0, ROOT)
    import mixdq_amd._C as C
    dev = "cuda:0"

    def timed(fn, L=400, reps=5):
        fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for i in range(L):
                fn(i)
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best = 1e9
        for _ in range(reps):
            e0.record()
            g.replay()
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e3 / L)
        return round(best, 2)

    gen = torch.Generator(device="cpu").manual_seed(0)
    one = torch.zeros(1, device=dev)
    x8 = torch.zeros(8, device=dev, dtype=torch.float16)
    s, z = torch.ones((), device=dev), torch.zeros((), device=dev)
    x = torch.randn(1024, 1280, generator=gen).half().to(dev)
    g1 = torch.ones(1280, device=dev, dtype=torch.float16)
    ws = [torch.randint(-128, 128, (1280, 1280), generator=gen, dtype=torch.int8).to(dev) for _ in range(40)]
    sc = (torch.rand(1280, generator=gen) * 1e-4).to(dev)
    a = torch.randint(-128, 128, (1024, 1280), generator=gen, dtype=torch.int8).to(dev)
    out = {}
    out["tiny"] = timed(lambda i=0: one.add_(1))
    out["quant8"] = timed(lambda i=0: C.quantize_per_tensor_to_int8(x8, s, z))
    out["ln"] = timed(lambda i=0: C.layernorm_quantize(x, g1, g1, 1e-5, [(s, z)]))
    out["gemm"] = timed(lambda i=0: C.qlinear_w8_a8_ohalf(a, ws[i % 40], sc, z, z, sc, sc, sc, None))

    def pair(i=0):
        q = C.layernorm_quantize(x, g1, g1, 1e-5, [(s, z)])[0][0]
        return C.qlinear_w8_a8_ohalf(q, ws[i % 40], sc, z, z, sc, sc, sc, None, _residual=x)
    out["pair"] = timed(pair, L=200)
    print(json.dumps(out))


if __name__ == "__main__":
    if "--sweep" in sys.argv:
        sweep()
    else:
        main()
