#!/bin/bash
# session 13: GroupNorm apply at small batch -- more, shorter blocks
out=gpurun_out/s13
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
for rep in 1 2; do
for d in 1 2 4 8; do
  echo "== mul $d" >> $out/norms.txt
  MIXDQ_GN_APPLY_MUL=$d timeout 300 python tools/bench_norms.py 2>&1 | grep "^GN (1\|^GN (2" >> $out/norms.txt
done
done
( time MIXDQ_GN_APPLY_MUL=4 timeout 900 python -m pytest tests/test_fused_gpu.py -q -m gpu -k "groupnorm or gn" 2>&1 | tail -4 ) > $out/pytest.txt 2>&1
cat $out/pytest.txt $out/norms.txt
