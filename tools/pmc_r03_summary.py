#!/usr/bin/env python3
"""<out>/raw/* (tools/pmc_r04.sh, round 3: tools/pmc_r03.sh) -> summary.json: per probed launch the HBM-side bytes
(FETCH_SIZE x 2 + WRITE_SIZE, KiB -> bytes: MI355X_MICROARCH.md, HBM section), the algorithmic bytes, and
from the SQ pass mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (4 x SQ_BUSY_CU_CYCLES) and the wave-cycle split."""
import collections
import csv
import glob
import json
import re
import sys

out = sys.argv[1]
res = {}
for line in open(out + "/shapes.txt"):
    i, kind, *v = line.split()
    v = [int(t) for t in v]
    peak_kind = "int8"
    if kind == "attn":
        B, T, Cc = v[:3]
        M, N, K = B * (Cc // 64) * T, T, 64                 # 4 M N K FLOPs: Q K^T and P V
        alg = 4 * 2 * B * T * Cc                            # q, k, v read + o written, fp16 / int8 mix: fp16 bound
        shape = f"attn B{B} T{T} C{Cc}"
        peak_kind = "f16"
    elif kind == "conv":
        NI, HW, CIN, COUT = v[:4]
        M, N, K = NI * HW * HW, COUT, 9 * CIN
        alg = NI * HW * HW * CIN + N * K + 2 * M * N
        shape = f"conv3x3 {NI}x{HW}x{HW}x{CIN}->{COUT}"
    elif kind == "linattn":
        B, T, Cc = v[:3]
        M, N, K = B * T, Cc, Cc                               # the to_q GEMM (the 77-key attention adds ~12 % FLOPs)
        alg = M * K + N * K + M * N + 2 * 2 * B * 77 * Cc     # int8 in, int8 out, k / v fp16
        shape = f"linattn M{M} N{N} K{K}"
    else:
        M, N, K = v[:3]
        alg = M * K + N * K + (M * N // 2 if kind == "geglu" else 2 * M * N)
        if kind == "ln":
            alg += 2 * M * N + M * N                          # residual read, one INT8 LayerNorm output written
        if kind == "f16in":
            alg += M * K                                      # the operand arrives as FP16
        shape = f"{kind} M{M} N{N} K{K}"
    entry = {}
    for p in ("FETCH_SIZE", "WRITE_SIZE", "SQ", "SQ2"):
        f = glob.glob(f"{out}/raw/{p}_{i}/**/*counter_collection.csv", recursive=True)
        if not f:
            continue
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f[0])):
            if not any(t in r["Kernel_Name"] for t in ("igemm_kernel", "igemm_pp_kernel", "conv3x3_halo", "attn_fwd_kernel", "attn_short_kernel")):
                continue
            k = re.sub(r"\(.*$", "", r["Kernel_Name"].replace("mixdq::(anonymous namespace)::", "").replace("void ", ""))
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            agg[k]["_dur_ns"].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        for k, d in agg.items():
            e = entry.setdefault(k, {})
            for c, vals in d.items():
                vals = vals[1:] or vals                       # first launch: cold instruction cache
                e[(p + ":" if c == "_dur_ns" else "") + c] = sum(vals) / len(vals)
    for k, e in entry.items():
        fk, wk = e.get("FETCH_SIZE", 0.0), e.get("WRITE_SIZE", 0.0)
        e["shape"] = shape
        e["algorithmic_bytes"] = alg
        e["hbm_bytes_per_launch"] = int((2 * fk + wk) * 1024)   # gfx950: FETCH_SIZE counts half
        e["traffic_over_algorithmic"] = e["hbm_bytes_per_launch"] / alg
        if "SQ_VALU_MFMA_BUSY_CYCLES" in e and e.get("SQ_BUSY_CU_CYCLES"):
            e["mfma_util"] = e["SQ_VALU_MFMA_BUSY_CYCLES"] / (4.0 * e["SQ_BUSY_CU_CYCLES"])
        if e.get("SQ_WAVE_CYCLES"):
            for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
                if c in e:
                    e[c + "_frac_of_wave_cycles"] = e[c] / e["SQ_WAVE_CYCLES"]
        dur = e.get("SQ:_dur_ns", e.get("FETCH_SIZE:_dur_ns", 1))
        if peak_kind == "f16":
            e["f16_tflops_profiled"] = 4.0 * M * N * K / (dur * 1e-9) / 1e12
        else:
            e["int8_tops_profiled"] = 2.0 * M * N * K / (dur * 1e-9) / 1e12
        # VALU-busy (second SQ pass): vector-ALU instruction cycles of all waves over the CUs' busy cycles x 4 SIMDs
        if "SQ_ACTIVE_INST_VALU" in e and e.get("SQ_BUSY_CU_CYCLES"):
            e["valu_util"] = 4.0 * e["SQ_ACTIVE_INST_VALU"] / (4.0 * e["SQ_BUSY_CU_CYCLES"])   # quad-cycles -> cycles
        res[f"{k} @ {shape}"] = e
json.dump(res, open(out + "/summary.json", "w"), indent=1, sort_keys=True)
for k, e in sorted(res.items()):
    print(k[:110], "| us", round(e.get("SQ:_dur_ns", 0) / 1e3, 1), "| hbm/alg", round(e["traffic_over_algorithmic"], 2),
          "| mfma util", round(e.get("mfma_util", 0), 3), "| wait", round(e.get("SQ_WAIT_ANY_frac_of_wave_cycles", 0), 2))
