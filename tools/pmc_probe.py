#!/usr/bin/env python3
"""Launch the hot-path kernels on the dominant SDXL shapes, a few times each, for rocprofv3 --pmc
passes (FETCH_SIZE / WRITE_SIZE in separate runs).  Inputs are made on the CPU and copied, and no
PyTorch GPU kernel is launched: rocprofv3 --pmc segfaults inside some torch reduction launches on
this image, so bench.py itself cannot be run under --pmc.

    rocprofv3 --pmc FETCH_SIZE --output-format csv -d out -- python3 tools/pmc_probe.py
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mixdq_amd._C as C  # noqa: E402

DEV = "cuda:0"
REPS = 5


def dev(t):
    return t.to(DEV)


def main():
    g = torch.Generator().manual_seed(0)
    zero = dev(torch.zeros(()))
    s_inv, zp = dev(torch.tensor(20.0)), dev(torch.tensor(3.0))
    # igemm, linear: (M, N, K)
    for M, N, K in [(1024, 1280, 1280), (1024, 10240, 1280), (1024, 1280, 5120), (4096, 640, 640),
                    (8192, 1280, 1280)]:
        a = dev(torch.randint(-128, 128, (M, K), generator=g, dtype=torch.int8))
        w = dev(torch.randint(-128, 128, (N, K), generator=g, dtype=torch.int8))
        sc = dev(torch.rand(N, generator=g) * 1e-4)
        b0 = dev(torch.rand(N, generator=g) * 100)
        for _ in range(REPS):
            C.qlinear_w8_a8_ohalf(a, w, sc, zero, zero, b0, sc, b0, None)
    # igemm, conv 3x3: (H, Cin, Cout)
    for hw, cin, cout in [(32, 1280, 1280), (128, 320, 320), (64, 640, 640)]:
        x = dev(torch.randint(-128, 128, (1, hw, hw, cin), generator=g, dtype=torch.int8)
                ).permute(0, 3, 1, 2)
        w = dev(torch.randint(-128, 128, (cout, 3, 3, cin), generator=g, dtype=torch.int8)
                ).permute(0, 3, 1, 2)
        wsum = dev(torch.randint(-128, 128, (cout, 3, 3, cin), generator=g, dtype=torch.int8
                                 ).float().sum(dim=3).reshape(cout, 1, 3, 3))
        table = C.conv_border_table(wsum)
        sc = dev(torch.rand(cout, generator=g) * 1e-4)
        for _ in range(REPS):
            C.qconv2d_w8_a8_ohalf(x, w, sc, zero, zp, sc, wsum, None, None, 1, 1, 1, _table=table)
    # quantize / fused producers
    for shape in [(1024, 1280), (16384, 320), (16384, 1920)]:
        x = dev(torch.randn(*shape, generator=g).half())
        for _ in range(REPS):
            C.quantize_per_tensor_to_int8(x, s_inv, zp)
    x = dev(torch.randn(1024, 1280, generator=g).half())
    gm = dev(torch.ones(1280).half())
    for _ in range(REPS):
        C.layernorm_quantize(x, gm, gm, 1e-5, [(s_inv, zp)])
    h = dev(torch.randn(1024, 10240, generator=g).half())
    for _ in range(REPS):
        C.geglu_quantize(h, s_inv, zp)
    x = dev(torch.randn(1, 128, 128, 320, generator=g).half()).permute(0, 3, 1, 2)
    gm = dev(torch.ones(320).half())
    for _ in range(REPS):
        C.groupnorm_silu_quantize(x, 32, gm, gm, 1e-5, s_inv, zp)
    torch.cuda.synchronize()
    print("probe done")


if __name__ == "__main__":
    main()
