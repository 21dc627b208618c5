#!/bin/bash
# A/B builds for the kernel-argument preload experiment:
#   build/kp/libmixdq_nokp.so  -DMIXDQ_KP=0, no preload flag (round-3 argument passing)
#   build/kp/libmixdq_kp.so    -DMIXDQ_KP=1 + -mllvm -amdgpu-kernarg-preload-count=14 on every file
# (measured, profiles/r04_kernarg_preload_ab.txt: igemm gains, the small kernels lose -> mixdq_amd/build.py sets
#  the flag for igemm.hip only)
set -e
cd "$(dirname "$0")/.."
mkdir -p build/kp
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-function"
KP="-mllvm -amdgpu-kernarg-preload-count=14"
for f in quantize igemm iconv fused_norm; do
  /opt/rocm/bin/hipcc $F -DMIXDQ_KP=0 -c -o build/kp/${f}_nokp.o mixdq_amd/csrc/$f.hip &
  /opt/rocm/bin/hipcc $F -DMIXDQ_KP=1 $KP -c -o build/kp/${f}_kp.o mixdq_amd/csrc/$f.hip &
done
/opt/rocm/bin/hipcc $F -mllvm -amdgpu-mfma-vgpr-form=1 -DMIXDQ_KP=0 -c -o build/kp/attention_nokp.o mixdq_amd/csrc/attention.hip &
/opt/rocm/bin/hipcc $F -mllvm -amdgpu-mfma-vgpr-form=1 -DMIXDQ_KP=1 $KP -c -o build/kp/attention_kp.o mixdq_amd/csrc/attention.hip &
wait
for v in nokp kp; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/kp/libmixdq_$v.so build/kp/quantize_$v.o build/kp/igemm_$v.o \
    build/kp/iconv_$v.o build/kp/fused_norm_$v.o build/kp/attention_$v.o
done
ls -la build/kp/*.so
