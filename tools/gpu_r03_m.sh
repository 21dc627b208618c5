#!/bin/bash
out=gpurun_out/r03_m
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
export MIXDQ_HIP_LIB=$PWD/build/stamp/libmixdq_stamp.so
for spec in "8192 10240 1280 --geglu --cfg 70"; do
  echo "== $spec" >> $out/stamps.txt
  timeout 300 python tools/stamp_report.py $spec 2>&1 | grep -v amdgpu.ids >> $out/stamps.txt
done
cat $out/stamps.txt
