#!/usr/bin/env python3
"""profiles/r04_pmc_summary.json (tools/pmc_r04.sh) -> profiles/pmc_traffic.json, the table bench.py
reads `roofline.traffic` / `roofline.mfma_util` from: keyed by bench.py's kernel names, one section per
batch size of the probed launch ("bs1", "bs8": the launch's rows / 1024 or its image count)."""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "r06_pmc_summary.json")
lease = sys.argv[2] if len(sys.argv) > 2 else "unknown"
lib_sha = sys.argv[3] if len(sys.argv) > 3 else None      # the hash embedded in the library the counters were taken on
tag = os.path.basename(src)
out = {"_comment": "HBM-side bytes per launch and MFMA-busy fraction from the rocprofv3 --pmc passes of "
                   "tools/pmc_r06.sh (profiles/r06_pmc_*): FETCH_SIZE x2 (gfx950 correction) + WRITE_SIZE, "
                   "KiB -> bytes, separate passes; mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (4 x "
                   "SQ_BUSY_CU_CYCLES).  Sections = batch size of the benchmarked graph, keys = bench.py "
                   "kernel names; each entry was measured on that very instantiation and launch kind."}
import hashlib   # noqa: E402
h = hashlib.sha256()
d = os.path.join(ROOT, "mixdq_amd", "csrc")
for name in sorted(os.listdir(d)):          # = bench.py csrc_sha16(): the kernel sources the counters were taken on
    if name.endswith((".hip", ".h")):
        with open(os.path.join(d, name), "rb") as f:
            h.update(name.encode() + b"\0" + f.read())
out["_provenance"] = {"summary": f"profiles/{tag}", "script": "tools/pmc_r06.sh", "lease": lease,
                      "csrc_sha16": lib_sha or h.hexdigest()[:16],
                      "csrc_sha16_from": "the probed library (mixdq_build_csrc_sha16)" if lib_sha else "the source tree at table-generation time",
                      "note": "collected by the builder's rocprofv3 --pmc passes; bench.py re-prints these columns and "
                              "drops them when csrc_sha16 differs from the running tree's"}
for key, e in json.load(open(src)).items():
    kern, shape = key.split(" @ ")
    args = [a.strip() for a in re.search(r"<(.*)>", kern).group(1).split(",")]
    if kern.startswith("attn_fwd_kernel"):
        m = re.search(r"attn B(\d+) T(\d+)", shape)
        name, bs = f"attn_fwd_kernel<Tkv={m.group(2)}>", int(m.group(1))
    elif kern.startswith("conv3x3_halo"):
        name = f"conv3x3_halo_kernel<{args[0]},{args[1]},{args[2]}>"
        bs = int(re.search(r"conv3x3 (\d+)x", shape).group(1))
    elif kern.startswith("igemm_pp_kernel"):       # the persistent four-phase kernel: configuration 71
        kind = "linear_geglu" if shape.split()[0] == "geglu" else "linear"
        name = f"igemm_kernel<256,256,128,2,{kind}>#cfg71"
        rows = int(re.search(r"M(\d+)", shape).group(1))
        ncols = int(re.search(r"N(\d+)", shape).group(1))
        bs = max(1, rows // (4096 if ncols in (640, 5120) and rows % 4096 == 0 else 1024))
    else:
        kind = {"geglu": "linear_geglu", "linattn": "linear_attn", "ln": "linear_ln", "f16in": "linear_f16in"}.get(
            shape.split()[0], "linear")
        name = f"igemm_kernel<{args[0]},{args[1]},{args[2]},{args[3]},{kind}>"
        if kind in ("linear_attn", "linear_ln", "linear_f16in"):
            name += "#cfg" + {("64", "128"): "41", ("64", "80"): "56" if args[3] == "6" else "45", ("128", "80"): "44"}.get(
                (args[0], args[1]), "0")
        if args[:4] == ["128", "320", "128", "2"]:      # one tile, three wave layouts: keep them apart
            name += {("4", "2"): "#cfg25", ("8", "2"): "#cfg27", ("4", "4"): "#cfg28"}.get((args[4], args[5]), "")
        rows = int(re.search(r"M(\d+)", shape).group(1))
        ncols = int(re.search(r"N(\d+)", shape).group(1))
        bs = max(1, rows // (4096 if ncols in (640, 5120) and rows % 4096 == 0 else 1024))   # rows per image: 4096 at 640 channels
    out.setdefault(f"bs{bs}", {})[name] = {
        "shape": shape, "hbm_bytes_per_launch": e["hbm_bytes_per_launch"],
        "algorithmic_bytes": e["algorithmic_bytes"],
        "traffic_over_algorithmic": round(e["traffic_over_algorithmic"], 3),
        "mfma_util": round(e.get("mfma_util", 0.0), 4),
        "us_under_pmc": round(e.get("SQ:_dur_ns", 0) / 1e3, 1),
        "source": f"profiles/{tag} (tools/pmc_r06.sh)"}
    if "valu_util" in e:
        out[f"bs{bs}"][name]["valu_util"] = round(e["valu_util"], 4)
dst = os.path.join(ROOT, "profiles", "pmc_traffic.json")
json.dump(out, open(dst, "w"), indent=1)
print(dst, {k: len(v) for k, v in out.items() if k != "_comment"})
