#!/bin/bash
# session 6 (s5 + stores through the nontemporal builtin, raw epilogue barriers, cfg 27 selected): late residual batch (fixed), batched epilogue-vector reads, short-key attention, 128x320 on 16 waves
out=gpurun_out/s6
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
A=$PWD/build/ab_r04a/libmixdq_hip.so
( time timeout 1800 python -m pytest tests -q -m gpu -k "not path_a" 2>&1 | tail -12 ) > $out/pytest.txt 2>&1
for v in A C A C; do
  lib=""; [ $v = A ] && lib=$A
  MIXDQ_HIP_LIB=$lib timeout 600 python bench.py --no-fp16 --no-cpu-baseline --no-roofline --no-batch8 --no-dropin --steps 50 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v bs1', d['ms_per_step'])" >> $out/bench.txt
done
for v in A C A C; do
  lib=""; [ $v = A ] && lib=$A
  MIXDQ_HIP_LIB=$lib timeout 600 python bench.py --no-fp16 --no-cpu-baseline --no-roofline --steps 20 --batch 8 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v bs8', d['ms_per_step'])" >> $out/bench.txt
done
timeout 600 python tools/bench_geglu_cfgs.py > $out/geglu_cfgs.txt 2>&1
timeout 300 python tools/bench_gemm.py --bs 8 --cfgs 70,25,27 2>&1 | grep -v amdgpu | cut -c1-220 > $out/gemm_bs8.txt
export MIXDQ_HIP_LIB=$PWD/build/stamp/libmixdq_stamp.so
for spec in "1024 10240 1280 --geglu --cfg 25" "8192 10240 1280 --geglu --cfg 70" "8192 1280 1280 --cfg 25 --res" "8192 1280 5120 --cfg 70 --res"; do
  echo "== $spec" >> $out/stamps.txt
  timeout 300 python tools/stamp_report.py $spec 2>&1 | grep -v amdgpu.ids | tail -2 >> $out/stamps.txt
done
unset MIXDQ_HIP_LIB
cat $out/pytest.txt $out/bench.txt $out/geglu_cfgs.txt $out/stamps.txt
