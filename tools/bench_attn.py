"""Microbenchmark of the FP16 attention core on the SDXL UNet's shapes (hipGraph-timed)."""
import argparse, json, sys, os
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

SHAPES = [  # (name, Tq, Tkv, C, calls per forward)
    ("self 64x64 C640", 4096, 4096, 640, 10),
    ("self 32x32 C1280", 1024, 1024, 1280, 60),
    ("cross 64x64 C640", 4096, 77, 640, 10),
    ("cross 32x32 C1280", 1024, 77, 1280, 60),
]


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    for _ in range(10):          # bring the clocks up
        g.replay()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(5):
        g.replay()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / (5 * reps)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--bs", type=int, default=1)
    ap.add_argument("--impl", default="torch,hip")
    ap.add_argument("--cfg", type=lambda x: int(x, 0), default=0)
    ap.add_argument("--only", default="", help="substring of the shape name (e.g. cross)")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    total = {}
    for name, tq, tkv, c, calls in SHAPES:
        if a.only not in name:
            continue
        h = c // 64
        torch.manual_seed(0)
        if tq == tkv:
            qkv = torch.randn(a.bs, tq, 3 * c, device=dev, dtype=torch.float16)
            q, k, v = qkv[..., :c], qkv[..., c:2 * c], qkv[..., 2 * c:]
        else:
            q = torch.randn(a.bs, tq, c, device=dev, dtype=torch.float16)
            kv = torch.randn(a.bs, tkv, 2 * c, device=dev, dtype=torch.float16)
            k, v = kv[..., :c], kv[..., c:]
        row = {"shape": name, "bs": a.bs, "gflop": 4 * a.bs * tq * tkv * c / 1e9}
        for impl in a.impl.split(","):
            if impl == "torch":
                def fn():
                    qq = q.unflatten(-1, (h, 64)).transpose(1, 2)
                    kk = k.unflatten(-1, (h, 64)).transpose(1, 2)
                    vv = v.unflatten(-1, (h, 64)).transpose(1, 2)
                    return F.scaled_dot_product_attention(qq, kk, vv).transpose(1, 2).reshape(a.bs, tq, c)
            else:
                try:
                    from mixdq_amd import _C
                    if not hasattr(_C, "attention_f16"):
                        continue
                except ImportError:
                    continue
                def fn():
                    return _C.attention_f16(q, k, v, h, _cfg=a.cfg)
            us = timed(fn)
            row[impl + "_us"] = round(us, 2)
            row[impl + "_tflops"] = round(row["gflop"] / us * 1e3)
            total[impl] = total.get(impl, 0.0) + us * calls
        print(json.dumps(row))
    print(json.dumps({"per_forward_ms": {k: round(v / 1e3, 3) for k, v in total.items()}}))


if __name__ == "__main__":
    main()
