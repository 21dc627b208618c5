#!/bin/bash
out=gpurun_out/r03_g
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
S=$PWD/build/stamp/libmixdq_stamp.so
{
for args in "1 640 640 --conv 64 --res" "1 320 320 --conv 128 --res" "1 1280 1280 --conv 32 --res" "1 1280 2560 --conv 32"; do
  echo "== conv images K=Cin N=Cout: $args"
  MIXDQ_HIP_LIB=$S timeout 300 python tools/stamp_report.py $args 2>&1 | grep -v amdgpu.ids | tail -2
done
} > $out/stamps.log
cat $out/stamps.log
