#!/bin/bash
out=gpurun_out/r03_k
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
for bs in 1 2 4; do
timeout 600 python tools/bench_gemm.py --conv --bs $bs --cfgs 90,91,92 2>&1 | grep "^{" | python -c "
import sys, json
for l in sys.stdin:
    r=json.loads(l); print('bs $bs', r['shape'], r['us'], 'auto', r['auto_us'])"
done
