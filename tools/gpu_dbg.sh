#!/bin/bash
out=gpurun_out/dbg
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
timeout 600 python tools/dbg_halo.py > $out/halo.txt 2>&1
MIXDQ_HIP_LIB=$PWD/build/ab_s4/libmixdq_hip.so timeout 600 python tools/dbg_halo.py > $out/halo_s4lib.txt 2>&1
cat $out/halo.txt; echo ===== s4 lib; cat $out/halo_s4lib.txt
