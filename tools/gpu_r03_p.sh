#!/bin/bash
out=gpurun_out/r03_p
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
python tools/bench_norms.py > $out/norms_auto.txt 2>&1
MIXDQ_LN_ROWS=1 python tools/bench_norms.py 2>&1 | grep "^LN" > $out/norms_rows1.txt
MIXDQ_LN_ROWS=2 python tools/bench_norms.py 2>&1 | grep "^LN" > $out/norms_rows2.txt
cat $out/norms_auto.txt $out/norms_rows1.txt $out/norms_rows2.txt
