#!/bin/bash
out=gpurun_out/r03_s
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
( time timeout 1500 python -m pytest tests/test_ops_gpu.py tests/test_fused_gpu.py tests/test_modules_gpu.py -q -x ) > $out/pytest.log 2>&1
tail -12 $out/pytest.log
timeout 600 python tools/bench_gemm.py --w4 --cfgs 4,35,37,41,45,56,46,13 2>&1 | grep "^{" | python -c "
import sys, json
for l in sys.stdin:
    r=json.loads(l); print('w4', r['shape'], r['us'], 'auto', r['auto_us'])"
timeout 600 python tools/bench_gemm.py --cfgs 45,56,25,35 2>&1 | grep "^{" | python -c "
import sys, json
for l in sys.stdin:
    r=json.loads(l); print('w8', r['shape'], 'auto', r['auto_us'])"
