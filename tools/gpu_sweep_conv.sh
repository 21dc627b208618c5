#!/bin/bash
# Tile-rule audit, convolutions: every UNet conv shape under the halo tiles and the implicit-GEMM tiles.
out=gpurun_out/sweep
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
for bs in 1 8; do
timeout 900 python tools/bench_gemm.py --conv --bs $bs --cfgs 90,91,92,13,25,35,44,45,56,41,3 2>&1 | grep -v amdgpu > $out/conv_bs$bs.txt
done
python - <<PY
import json
for f in ("$out/conv_bs1.txt", "$out/conv_bs8.txt"):
    print("==", f)
    for l in open(f):
        if l.startswith("{"):
            r = json.loads(l)
            us = {k: v for k, v in r["us"].items() if not isinstance(v, str)}
            b = min(us, key=us.get)
            flag = "  <<<" if us[b] < 0.95 * r["auto_us"] else ""
            print(r["shape"], "x", r["count"], "auto", r["auto_us"], "best", b, us[b], {k: v for k, v in r["us"].items() if k in ("90", "91", "92")}, flag)
        else:
            print(l.strip())
PY
