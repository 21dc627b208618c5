#!/bin/bash
# Build the HIP library of another commit for same-box A/B runs: build/ab_<name>/libmixdq_hip.so
#   bash tools/ab_build.sh <commit> <name>;  MIXDQ_HIP_LIB=$PWD/build/ab_<name>/libmixdq_hip.so python bench.py ...
set -e
cd "$(dirname "$0")/.."
c=$1; n=$2; d=build/ab_$n
mkdir -p $d/mixdq_amd/csrc $d/include
for f in quantize.hip igemm.hip fused_norm.hip attention.hip common.h attn_core.h; do git show $c:mixdq_amd/csrc/$f > $d/mixdq_amd/csrc/$f; done
for f in mixdq_hip.h mixdq_math.h; do git show $c:include/$f > $d/include/$f; done
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-function"
pids=""
for f in quantize igemm fused_norm; do /opt/rocm/bin/hipcc $F -c -o $d/$f.o $d/mixdq_amd/csrc/$f.hip & pids="$pids $!"; done
/opt/rocm/bin/hipcc $F -mllvm -amdgpu-mfma-vgpr-form=1 -c -o $d/attention.o $d/mixdq_amd/csrc/attention.hip & pids="$pids $!"
for p in $pids; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $d/libmixdq_hip.so $d/quantize.o $d/igemm.o $d/fused_norm.o $d/attention.o
rm -rf $d/mixdq_amd $d/include $d/*.o
echo $d/libmixdq_hip.so
