#!/usr/bin/env python3
"""Per-phase time line of the self-attention launch from in-kernel stamps (diagnostic build, tools/stamp_build.sh):

    MIXDQ_HIP_LIB=$PWD/build/stamp/libmixdq_stamp.so python tools/stamp_attn.py B T C [--payload MB]

Slots: 0 entry (attention workgroups) | 1 Q and the prologue's K/V tiles requested | 2 first K/V tile landed |
3 Q in registers, scores of tile 0 | 4 tile loop done | 5 O normalised and staged in LDS | 7 rows stored.
Per slot: the median over workgroups of the FIRST and the LAST wave to reach it, us from the workgroup's own
entry (shader clocks of different XCDs are not synchronised).  Also: when the workgroups ENTER, from the real-time
counter (100 MHz, chip-wide): the spread of slot-0 real times over the launch = the dispatch ramp."""
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
    ap.add_argument("B", type=int)
    ap.add_argument("T", type=int)
    ap.add_argument("C", type=int)
    ap.add_argument("--payload", type=float, default=0.0, help="MB of prefetch payload riding on the launch")
    a = ap.parse_args()
    lib = C._lib
    assert hasattr(lib, "mixdq_debug_stamps_attn"), "not a stamped build (tools/stamp_build.sh)"
    lib.mixdq_debug_stamps_attn.argtypes = [ctypes.c_void_p]
    g = torch.Generator().manual_seed(0)
    qkv = torch.randn(a.B, a.T, 3 * a.C, generator=g).half().to(DEV)
    s_inv, z = torch.full((), 20.0, device=DEV), torch.zeros((), device=DEV)
    pay = [torch.empty(int(a.payload * 1e6), dtype=torch.int8, device=DEV)] if a.payload else None
    flush = torch.empty(512 << 20, dtype=torch.uint8, device=DEV)
    stamps = torch.zeros((1 << 15, 4, 16), dtype=torch.int64, device=DEV)

    def launch():
        C.attention_f16(qkv[..., :a.C], qkv[..., a.C:2 * a.C], qkv[..., 2 * a.C:], a.C // 64, s_inv, z, _prefetch=pay)

    for _ in range(3):
        launch()
    torch.cuda.synchronize()
    for rep in range(3):
        flush.zero_()                     # q / k / v come from the memory side, as behind the q|k|v GEMM
        stamps.zero_()
        torch.cuda.synchronize()
        lib.mixdq_debug_stamps_attn(ctypes.c_void_p(stamps.data_ptr()))
        launch()
        torch.cuda.synchronize()
        lib.mixdq_debug_stamps_attn(None)
        s = stamps.cpu().numpy().astype(np.int64)
        wg = np.nonzero((s[:, :, 0] != 0).any(axis=1))[0]
        s = s[wg]
        used = s[:, :, 0] != 0
        big = np.iinfo(np.int64).max
        t0 = np.where(used, s[:, :, 0], big).min(axis=1)
        dt_clk = (s[:, :, 7] - s[:, :, 0])[used]
        dt_rt = (s[:, :, 9] - s[:, :, 8])[used]
        ok = dt_rt > 0
        ghz = float(np.median(dt_clk[ok] / dt_rt[ok])) * 0.1 if ok.any() else 2.0
        parts = []
        for slot in (1, 2, 3, 4, 5, 7):
            v = s[:, :, slot]
            have = used & (v != 0)
            first = np.where(have, v, big).min(axis=1) - t0
            last = np.where(have, v, 0).max(axis=1) - t0
            parts.append(f"s{slot}: {np.median(first) / (ghz * 1e3):5.2f}..{np.median(last) / (ghz * 1e3):5.2f}")
        rt0 = np.where(used, s[:, :, 8], big).min(axis=1)           # real time (10 ns ticks) of each workgroup's entry
        rt7 = np.where(used, s[:, :, 9], 0).max(axis=1)
        print(f"clock {ghz:.2f} GHz, {len(wg)} attention workgroups | " + " | ".join(parts) +
              f" | entries spread over {(rt0.max() - rt0.min()) * 0.01:.2f} us, first entry -> last exit {(rt7.max() - rt0.min()) * 0.01:.2f} us")


if __name__ == "__main__":
    main()
