#!/bin/bash
# Round 6: the persistent kernel with the next tile's SECOND K-tile requested from the GEGLU epilogue -- tests, per-launch
# and whole-step A/B against the previous commit's library (build/ab_base)  -> gpurun_out/r06_pp2/
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
out=gpurun_out/r06_pp2
rm -rf $out; mkdir -p $out
timeout 900 python -m pytest tests/test_fused_gpu.py tests/test_ops_gpu.py tests/test_large_gpu.py -x -q -m gpu -k "persistent or 71 or geglu" > $out/pytest.txt 2>&1
tail -5 $out/pytest.txt
for v in base new base new; do
  lib=$PWD/mixdq_amd/libmixdq_hip.so; [ $v = base ] && lib=$PWD/build/ab_base/libmixdq_hip.so
  echo "== $v" >> $out/bench_pp.txt
  MIXDQ_HIP_LIB=$lib timeout 600 python tools/bench_pp.py 2>&1 | grep -v amdgpu.ids | grep geglu >> $out/bench_pp.txt
done
cat $out/bench_pp.txt
for rep in 1 2; do for v in base new; do
  lib=$PWD/mixdq_amd/libmixdq_hip.so; [ $v = base ] && lib=$PWD/build/ab_base/libmixdq_hip.so
  MIXDQ_HIP_LIB=$lib timeout 900 python bench.py --batch 8 --no-fp16 --no-cpu-baseline --no-roofline --no-dropin --no-lnchain --no-batch8 --steps 20 > $out/bench_bs8_${v}_$rep.json 2> $out/bench_bs8_${v}_$rep.err
  python3 - $out/bench_bs8_${v}_$rep.json $v $rep <<'PY' | tee -a $out/step_ab.txt
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], "rep", sys.argv[3], "batch 8 ms %.3f" % d["ms_per_step"])
except Exception as e:
    print(sys.argv[2], "ERR", e)
PY
done; done
