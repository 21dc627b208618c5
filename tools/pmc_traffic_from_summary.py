#!/usr/bin/env python3
"""profiles/r04_pmc_summary.json (tools/pmc_r04.sh) -> profiles/pmc_traffic.json, the table bench.py
reads `roofline.traffic` / `roofline.mfma_util` from: keyed by bench.py's kernel names, one section per
batch size of the probed launch ("bs1", "bs8": the launch's rows / 1024 or its image count)."""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "r04_pmc_summary.json")
tag = os.path.basename(src)
out = {"_comment": "HBM-side bytes per launch and MFMA-busy fraction from the rocprofv3 --pmc passes of "
                   "tools/pmc_r04.sh (profiles/r04_pmc_*): FETCH_SIZE x2 (gfx950 correction) + WRITE_SIZE, "
                   "KiB -> bytes, separate passes; mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (4 x "
                   "SQ_BUSY_CU_CYCLES).  Sections = batch size of the benchmarked graph, keys = bench.py "
                   "kernel names; each entry was measured on that very instantiation and launch kind."}
for key, e in json.load(open(src)).items():
    kern, shape = key.split(" @ ")
    args = [a.strip() for a in re.search(r"<(.*)>", kern).group(1).split(",")]
    if kern.startswith("attn_fwd_kernel"):
        m = re.search(r"attn B(\d+) T(\d+)", shape)
        name, bs = f"attn_fwd_kernel<Tkv={m.group(2)}>", int(m.group(1))
    elif kern.startswith("conv3x3_halo"):
        name = f"conv3x3_halo_kernel<{args[0]},{args[1]},{args[2]}>"
        bs = int(re.search(r"conv3x3 (\d+)x", shape).group(1))
    else:
        kind = "linear_geglu" if shape.startswith("geglu") else "linear"
        name = f"igemm_kernel<{args[0]},{args[1]},{args[2]},{args[3]},{kind}>"
        if args[:4] == ["128", "320", "128", "2"]:      # one tile, three wave layouts: keep them apart
            name += {("4", "2"): "#cfg25", ("8", "2"): "#cfg27", ("4", "4"): "#cfg28"}.get((args[4], args[5]), "")
        bs = max(1, int(re.search(r"M(\d+)", shape).group(1)) // 1024)
    out.setdefault(f"bs{bs}", {})[name] = {
        "shape": shape, "hbm_bytes_per_launch": e["hbm_bytes_per_launch"],
        "algorithmic_bytes": e["algorithmic_bytes"],
        "traffic_over_algorithmic": round(e["traffic_over_algorithmic"], 3),
        "mfma_util": round(e.get("mfma_util", 0.0), 4),
        "us_under_pmc": round(e.get("SQ:_dur_ns", 0) / 1e3, 1),
        "source": f"profiles/{tag} (tools/pmc_r04.sh)"}
    if "valu_util" in e:
        out[f"bs{bs}"][name]["valu_util"] = round(e["valu_util"], 4)
dst = os.path.join(ROOT, "profiles", "pmc_traffic.json")
json.dump(out, open(dst, "w"), indent=1)
print(dst, {k: len(v) for k, v in out.items() if k != "_comment"})
