#!/bin/bash
out=gpurun_out/r03_d
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
S=$PWD/build/stamp/libmixdq_stamp.so
{
for args in "1024 10240 1280 --geglu" "1024 1280 1280 --res" "1024 1280 5120 --res" "1024 3840 1280" "8192 10240 1280 --geglu"; do
  echo "== $args"
  MIXDQ_HIP_LIB=$S timeout 300 python tools/stamp_report.py $args 2>&1 | grep -v amdgpu.ids
done
} > $out/stamps.log
cat $out/stamps.log
( time timeout 2400 python -m pytest tests -m gpu -q -x ) > $out/pytest.log 2>&1
tail -8 $out/pytest.log
timeout 900 python bench.py --no-fp16 --no-cpu-baseline > $out/bench_bs1.json 2> $out/bench_bs1.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03_d/bench_bs1.json').read().strip().splitlines()[-1])
print('ms_per_step', d['ms_per_step'])
r=d['roofline']
print('dominant', r['kernel'], r['avg_launch_us'], r['frac'])
for k,v in sorted(r['per_kernel'].items(), key=lambda kv:-kv[1]['ms_per_step'])[:12]: print(round(v['ms_per_step'],3), v['launches'], round(v['tops']), k)
PY
timeout 900 python bench.py --no-fp16 --no-cpu-baseline --no-roofline --batch 8 --steps 10 > $out/bench_bs8.json 2> $out/bench_bs8.err
python -c "
import json
d=json.loads(open('gpurun_out/r03_d/bench_bs8.json').read().strip().splitlines()[-1]); print('bs8 ms_per_step', d['ms_per_step'])"
