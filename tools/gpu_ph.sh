#!/bin/bash
for v in base nt2 base nt2; do
  lib=$PWD/build/ab/libmixdq_$v.so; [ $v = base ] && lib=$PWD/mixdq_amd/libmixdq_hip.so
  echo "== $v"
  MIXDQ_HIP_LIB=$lib timeout 600 python tools/bench_gemm.py --bs 8 --cfgs 13,25,70 2>/dev/null | python3 -c "
import sys, json
for line in sys.stdin:
    if not line.startswith('{'): continue
    r = json.loads(line)
    if r['gops'] > 50: print('  ', r['shape'], r['us'])
"
done
MIXDQ_HIP_LIB=$PWD/build/ab/libmixdq_nt2.so bash tools/pmc_l2.sh 8192 10240 1280 13,70 2>&1 | grep -A4 "igemm_kernel<25"
