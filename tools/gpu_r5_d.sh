#!/bin/bash
# round 5, lease D: the whole GPU suite on the current tree + the default bench line
out=gpurun_out/r05_d
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
( time timeout 2400 python -m pytest tests -q -m gpu -x 2>&1 | tail -12 ) > $out/pytest_gpu.txt 2>&1
cat $out/pytest_gpu.txt
timeout 1500 python bench.py > $out/bench_default.json 2> $out/bench_default.err
tail -c 1500 $out/bench_default.err
python - <<PY
import json
d = json.loads(open("$out/bench_default.json").read().strip().splitlines()[-1])
r = d["roofline"]
print({k: d.get(k) for k in ("ms_per_step","speedup_vs_fp16","speedup_vs_fp16_like_for_like","speedup_vs_fp16_dropin","dropin_unfused_ms_per_step","dropin_unfused_kernels_per_step","kernels_per_step","multi_gpu")})
print({k: r.get(k) for k in ("kernel","frac","avg_launch_us","in_step_avg_launch_us","frac_in_step","in_step_launches","in_step_kernel_time_ms","in_step_kernels","traffic","traffic_source")})
print(d["batch8"]["ms_per_step"], d["cpu_baseline"]["value"])
PY
