#!/bin/bash
# Tile-rule audit: every UNet GEMM / conv shape under every tile configuration vs the automatic choice.
out=gpurun_out/sweep
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
timeout 900 python tools/bench_gemm.py --bs 1 2>&1 | grep -v amdgpu > $out/w8_bs1.txt
timeout 900 python tools/bench_gemm.py --bs 8 --cfgs 3,13,14,15,18,20,25,35,41,44,46,70 2>&1 | grep -v amdgpu > $out/w8_bs8.txt
timeout 900 python tools/bench_gemm.py --bs 2 --cfgs 3,4,13,18,25,35,37,41,42,43,44,45,56,46,70 2>&1 | grep -v amdgpu > $out/w8_bs2.txt
python - <<PY
import json
for f in ("$out/w8_bs1.txt", "$out/w8_bs2.txt", "$out/w8_bs8.txt"):
    print("==", f)
    for l in open(f):
        if l.startswith("{"):
            r = json.loads(l)
            us = {k: v for k, v in r["us"].items() if not isinstance(v, str)}
            b = min(us, key=us.get)
            flag = "  <<<" if us[b] < 0.95 * r["auto_us"] else ""
            print(r["shape"], "x", r["count"], "auto", r["auto_us"], "best", b, us[b], flag)
        else:
            print(l.strip())
PY
