#!/bin/bash
# Round 6: swap_glue tests + the default bench line with the dropin_glue legs -> gpurun_out/r06_glue/
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
out=gpurun_out/r06_glue
rm -rf $out; mkdir -p $out
timeout 900 python -m pytest tests/test_glue_gpu.py -x -q -m gpu > $out/pytest.txt 2>&1
tail -15 $out/pytest.txt
timeout 1500 python bench.py > $out/bench_bs1.json 2> $out/bench_bs1.err
tail -3 $out/bench_bs1.err
python3 - $out/bench_bs1.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
for k in ("ms_per_step", "dropin_unfused_ms_per_step", "dropin_glue_torch_sdpa_ms_per_step", "dropin_glue_ms_per_step",
          "dropin_unfused_kernels_per_step", "dropin_glue_kernels_per_step", "kernels_per_step", "speedup_vs_fp16",
          "speedup_vs_fp16_dropin", "speedup_vs_fp16_dropin_glue", "speedup_vs_fp16_dropin_glue_torch_sdpa",
          "speedup_vs_fp16_like_for_like", "dropin_glue_swapped_modules"):
    print(k, d.get(k))
print("fp16", d.get("fp16"))
print("batch8", d["batch8"]["ms_per_step"], d["batch8"].get("whole_step_frac"))
r = d["roofline"]
print({k: r.get(k) for k in ("kernel", "frac", "avg_launch_us", "whole_step_frac", "frac_in_step", "rocprof_avg_launch_us")})
PY
