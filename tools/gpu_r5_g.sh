#!/bin/bash
out=gpurun_out/r05_g
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
export MIXDQ_HIP_LIB=$PWD/build/stamp/libmixdq_stamp.so
for shape in "1024 1280 1280" "1024 1280 5120" "4096 640 640"; do
  echo "== $shape --res (plain GEMM)"; timeout 300 python tools/stamp_report.py $shape --res 2>&1 | tail -2
  echo "== $shape --res --ln"; timeout 300 python tools/stamp_report.py $shape --res --ln 2>&1 | tail -2
done > $out/stamps_ln.txt 2>&1
cat $out/stamps_ln.txt
