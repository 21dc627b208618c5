#!/bin/bash
# Packed-W4 launches of the batch-1 / batch-8 step under forced tile configurations (plain Linear).
out=gpurun_out/w4geglu
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
timeout 600 python tools/bench_gemm.py --w4 --bs 1 --cfgs 3,4,35,37,41,25,46,13,18,20 2>&1 | grep -v amdgpu > $out/w4_bs1.txt
timeout 600 python tools/bench_gemm.py --w4 --bs 8 --cfgs 3,35,25,46,13,18,20 2>&1 | grep -v amdgpu > $out/w4_bs8.txt
python - <<PY
import json
for f in ("$out/w4_bs1.txt", "$out/w4_bs8.txt"):
    for l in open(f):
        if l.startswith("{"):
            r = json.loads(l); print(r["shape"], r["us"], "auto", r["auto_us"])
        else:
            print(l.strip())
PY
