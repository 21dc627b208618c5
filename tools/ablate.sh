#!/bin/bash
# Ablation timing of the igemm main loop (diagnostic builds build/ab/libmixdq_ab{0..3}.so made with
# -DMIXDQ_ABLATE=n: 0 full, 1 no MFMA, 2 no LDS fragment reads, 3 no LDS-DMA in the loop, 4 no output stores).
CF=${1:-13,20,14,15,25,35}
BS=${2:-8}
for v in 0 1 2 3 4; do
  echo "== ABLATE=$v (0 full, 1 no MFMA, 2 no LDS reads, 3 no DMA, 4 no stores)"
  MIXDQ_HIP_LIB=$PWD/build/ab/libmixdq_ab$v.so timeout 600 python tools/bench_gemm.py --bs $BS --cfgs $CF 2>/dev/null | python3 -c "
import sys, json
for line in sys.stdin:
    if not line.startswith('{'): continue
    r = json.loads(line)
    if r['shape'] not in ('lin M8192 N10240 K1280', 'lin M8192 N1280 K5120', 'lin M1024 N1280 K1280', 'lin M1024 N10240 K1280', 'lin M1024 N1280 K5120'): continue
    print(' ', r['shape'], {k: (v if not isinstance(v, str) else float(v[9:-1])) for k, v in r['us'].items()})
"
done
