#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
out=gpurun_out/r06_pp_stamps2
rm -rf $out; mkdir -p $out
bash tools/stamp_build.sh > $out/build.log 2>&1 || tail -5 $out/build.log
export MIXDQ_HIP_LIB=$PWD/build/stamp/libmixdq_stamp.so
{
echo "== (8192, 10240, 1280) GEMM+GEGLU persistent (cfg 71), gates +-20"; timeout 300 python tools/stamp_report.py 8192 10240 1280 --geglu --cfg 71 --pp 2>&1 | grep clock
echo "== cfg 70"; timeout 300 python tools/stamp_report.py 8192 10240 1280 --geglu --cfg 70 2>&1 | grep clock
} > $out/stamps.txt
cat $out/stamps.txt
