#!/bin/bash
out=gpurun_out/r03_f
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
( time timeout 2400 python -m pytest tests/test_ops_gpu.py tests/test_modules_gpu.py tests/test_fused_gpu.py tests/test_f16_gpu.py tests/test_large_gpu.py -q -x ) > $out/pytest.log 2>&1
tail -5 $out/pytest.log
S=$PWD/build/stamp/libmixdq_stamp.so
{
for args in "1024 10240 1280 --geglu" "1024 1280 1280 --res" "1024 1280 5120 --res" "1024 3840 1280"; do
  echo "== $args"
  MIXDQ_HIP_LIB=$S timeout 300 python tools/stamp_report.py $args 2>&1 | grep -v amdgpu.ids | tail -2
done
} > $out/stamps.log
cat $out/stamps.log
for lib in r02 cur; do
  L=$PWD/mixdq_amd/libmixdq_hip.so; [ $lib = r02 ] && L=$PWD/build/ab_r02/libmixdq_hip.so
  for bs in 1; do
    MIXDQ_HIP_LIB=$L timeout 900 python bench.py --no-fp16 --no-cpu-baseline --batch $bs --steps 20 > $out/bench_${lib}_bs${bs}.json 2> $out/bench_${lib}_bs${bs}.err
    python - $out/bench_${lib}_bs${bs}.json $lib $bs <<'PY'
import json, sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r=d['roofline']
print(sys.argv[2], 'bs', sys.argv[3], 'ms_per_step %.3f' % d['ms_per_step'], '| igemm ms %.3f' % r['all_igemm']['ms_per_step'], '| dominant', r['kernel'], '%.1f us' % r['avg_launch_us'])
PY
  done
done
