#!/usr/bin/env python3
"""Per-phase time line of one igemm launch from in-kernel stamps (diagnostic build, tools/stamp_build.sh):

    MIXDQ_HIP_LIB=$PWD/build/stamp/libmixdq_stamp.so python tools/stamp_report.py M N K [--geglu] [--cfg id] [--res]

Slots: 0 entry | 12 first prologue stage issued | 13 rest of the argument block arrived, epilogue operands requested | 1 prologue DMA issued | 2 first K-tile landed | 3 main loop done (this wave) | 4 past
the barrier behind the main loop | 5 accumulators -> fp16 tile in LDS (this wave) | 6 past the barrier
behind it | 7 stores / GEGLU done | 10, 11 (register GEGLU of the 256x256 tile): INT8 tile written, past
the barrier behind it.  The shader clocks of different XCDs are not synchronised, so every
time is taken relative to the workgroup's own earliest entry stamp; printed per slot: the median over
workgroups of the FIRST and of the LAST wave to reach it, in us (shader clock / 100 MHz real-time)."""
import argparse
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mixdq_amd._C as C  # noqa: E402

DEV = "cuda:0"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("M", type=int)
    ap.add_argument("N", type=int)
    ap.add_argument("K", type=int)
    ap.add_argument("--geglu", action="store_true")
    ap.add_argument("--res", action="store_true")
    ap.add_argument("--ln", action="store_true", help="the GEMM + residual + LayerNorm + quantize launch (slots 6: tile "
                    "final | 10: records + output rows issued | 11: all records in | 14: row statistics done | 15: end)")
    ap.add_argument("--pp", action="store_true", help="the persistent four-phase kernel (--cfg 71; csrc/igemm_pp.h): "
                    "FIRST tile 1 prologue issued | 2 first K-tile ready | 3 main loop done | 5 accumulators -> fp16 | "
                    "6 table landed (plain: first pass staged) | 7 epilogue done;  SECOND tile 10 past its first barrier "
                    "(its first K-tile landed under the epilogue) | 11 main loop done | 12 accumulators -> fp16 | 13 table landed | "
                    "15 first store pass done | 14 epilogue done")
    ap.add_argument("--cfg", type=int, default=0)
    ap.add_argument("--cold", action="store_true", help="stream 512 MB between launches")
    ap.add_argument("--conv", type=int, default=0, metavar="HW",
                    help="3x3 pad-1 conv on an HW x HW image instead: M = batch rows (images), K = Cin, N = Cout")
    a = ap.parse_args()
    lib = C._lib
    assert hasattr(lib, "mixdq_debug_stamps"), "not a stamped build (tools/stamp_build.sh)"
    lib.mixdq_debug_stamps.argtypes = [ctypes.c_void_p]
    set_stamps = lib.mixdq_debug_stamps
    if a.ln:
        lib.mixdq_debug_stamps_ln.argtypes = [ctypes.c_void_p]
        set_stamps = lib.mixdq_debug_stamps_ln
        gm, bt = torch.ones(a.N, device=DEV).half(), torch.zeros(a.N, device=DEV).half()
        lnws = C.qlinear_ln_workspace(a.M, a.N, DEV)
    g = torch.Generator().manual_seed(0)
    x = torch.randint(-128, 128, (a.M, a.K), generator=g, dtype=torch.int8).to(DEV)
    w = torch.randint(-128, 128, (a.N, a.K), generator=g, dtype=torch.int8).to(DEV)
    sc = (torch.rand(a.N, generator=g) * 1e-4).to(DEV)
    one, z = torch.ones((), device=DEV), torch.zeros((), device=DEV)
    res = torch.randn(a.M, a.N, generator=g).half().to(DEV) if a.res else None
    flush = torch.empty(512 << 20, dtype=torch.uint8, device=DEV)
    grid_max = 1 << 16
    stamps = torch.zeros((grid_max, 16, 16), dtype=torch.int64, device=DEV)

    if a.conv:
        xc = torch.randint(-128, 128, (a.M, a.conv, a.conv, a.K), generator=g, dtype=torch.int8).to(DEV).permute(0, 3, 1, 2)
        wc = torch.randint(-128, 128, (a.N, 3, 3, a.K), generator=g, dtype=torch.int8).to(DEV).permute(0, 3, 1, 2)
        wsum = wc.float().sum(dim=1, keepdim=True)
        table = C.conv_border_table(wsum)
        resc = torch.randn(a.M, a.conv, a.conv, a.N, generator=g).half().to(DEV).permute(0, 3, 1, 2) if a.res else None

    def launch():
        if a.conv:
            C.qconv2d_w8_a8_ohalf(xc, wc, sc, z, one, sc, wsum, None, None, 1, 1, 1, _table=table,
                                  _cfg=a.cfg, _residual=resc)
        elif a.ln:
            C.qlinear_ln(x, w, sc, sc, None, res, gm, bt, 1e-5, [(one, z)], lnws, _cfg=a.cfg)
        elif a.geglu:
            C.qlinear_geglu(x, w, sc, sc, None, one, z, _cfg=a.cfg)
        else:
            C.qlinear_w8_a8_ohalf(x, w, sc, z, z, sc, sc, sc, None, _cfg=a.cfg, _residual=res)

    for _ in range(3):
        launch()
    torch.cuda.synchronize()
    for rep in range(4):
        if a.cold:
            flush.zero_()
        stamps.zero_()
        torch.cuda.synchronize()
        set_stamps(ctypes.c_void_p(stamps.data_ptr()))
        launch()
        torch.cuda.synchronize()
        set_stamps(None)
        s = stamps.cpu().numpy().astype(np.int64)
        wg = np.nonzero((s[:, :, 0] != 0).any(axis=1))[0]
        s = s[wg]
        used = s[:, :, 0] != 0                                  # [wg, wave]
        big = np.iinfo(np.int64).max
        t0 = np.where(used, s[:, :, 0], big).min(axis=1)        # per workgroup
        dt_clk = (s[:, :, 7] - s[:, :, 0])[used]
        dt_rt = (s[:, :, 9] - s[:, :, 8])[used]
        ok = dt_rt > 0
        ghz = float(np.median(dt_clk[ok] / dt_rt[ok])) * 0.1 if ok.any() else 2.0
        parts = []
        end = 15 if a.ln else 14 if a.pp else 7
        order = (12, 13, 1, 2, 3, 4, 5, 7, 6, 10, 11, 14, 15) if a.ln else (12, 13, 1, 2, 3, 4, 5, 6, 10, 11, 7)
        if a.pp:
            order = (1, 2, 3, 5, 6, 7, 10, 11, 12, 13, 15, 14)
        for slot in order:
            v = s[:, :, slot]
            have = used & (v != 0)
            if not have.any():
                continue
            first = np.where(have, v, big).min(axis=1) - t0
            last = np.where(have, v, 0).max(axis=1) - t0
            sel = have.any(axis=1)
            parts.append(f"s{slot}: {np.median(first[sel]) / (ghz * 1e3):5.2f}..{np.median(last[sel]) / (ghz * 1e3):5.2f}")
        span = (np.where(used, s[:, :, end], 0).max(axis=1) - t0) / (ghz * 1e3)
        print(f"clock {ghz:.2f} GHz, {len(wg)} workgroups, workgroup life med {np.median(span):.2f} max {span.max():.2f} us | "
              + " | ".join(parts))


if __name__ == "__main__":
    main()
