#!/bin/bash
# session 2: kernel-argument preload A/B (build/kp/libmixdq_{nokp,kp}.so)
out=gpurun_out/s2
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
for v in nokp kp nokp kp; do
  MIXDQ_HIP_LIB=$PWD/build/kp/libmixdq_$v.so timeout 300 python tools/floor_probe.py 2>&1 | grep "^{" | sed "s/^/$v /" >> $out/floor.txt
done
MIXDQ_HIP_LIB=$PWD/build/kp/libmixdq_kp.so timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_fused_gpu.py -q -m gpu -x 2>&1 | tail -4 > $out/pytest_kp.txt
for v in nokp kp nokp kp; do
  MIXDQ_HIP_LIB=$PWD/build/kp/libmixdq_$v.so timeout 600 python bench.py --no-fp16 --no-cpu-baseline --no-roofline --steps 50 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v bs1', d['ms_per_step'])" >> $out/bench.txt
done
for v in nokp kp; do
  MIXDQ_HIP_LIB=$PWD/build/kp/libmixdq_$v.so timeout 600 python bench.py --no-fp16 --no-cpu-baseline --no-roofline --steps 20 --batch 8 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v bs8', d['ms_per_step'])" >> $out/bench.txt
done
cat $out/floor.txt $out/pytest_kp.txt $out/bench.txt
