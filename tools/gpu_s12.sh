#!/bin/bash
# session 12: GroupNorm stats / apply -- next pixel requested one iteration ahead; fewer, longer apply blocks
out=gpurun_out/s12
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
base=$PWD/build/ab_base/libmixdq_hip.so
( time timeout 900 python -m pytest tests/test_fused_gpu.py -q -m gpu -k "groupnorm or gn" 2>&1 | tail -4 ) > $out/pytest.txt 2>&1
for rep in 1 2; do
echo "== base" >> $out/norms.txt
MIXDQ_HIP_LIB=$base timeout 300 python tools/bench_norms.py 2>&1 | grep "^GN" >> $out/norms.txt
for d in 1 2 4 8; do
  echo "== new div $d" >> $out/norms.txt
  MIXDQ_GN_APPLY_DIV=$d timeout 300 python tools/bench_norms.py 2>&1 | grep "^GN" >> $out/norms.txt
done
done
cat $out/pytest.txt $out/norms.txt
