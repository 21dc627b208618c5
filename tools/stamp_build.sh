#!/bin/bash
# Diagnostic build with in-kernel phase stamps (csrc/igemm.hip MIXDQ_STAMP): build/stamp/libmixdq_stamp.so.
# Use with  MIXDQ_HIP_LIB=$PWD/build/stamp/libmixdq_stamp.so python tools/stamp_report.py ...
set -e
# (igemm_aq.o is taken from mixdq_amd/_obj: the object mixdq_amd/build.py compiled AND ran tools/check_aq_isa.py on)
cd "$(dirname "$0")/.."
mkdir -p build/stamp
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-function"
/opt/rocm/bin/hipcc $F -mllvm -amdgpu-kernarg-preload-count=14 -DMIXDQ_STAMP=1 -c -o build/stamp/igemm.o mixdq_amd/csrc/igemm.hip
/opt/rocm/bin/hipcc $F -mllvm -amdgpu-kernarg-preload-count=14 -DMIXDQ_STAMP=1 -c -o build/stamp/igemm_ln.o mixdq_amd/csrc/igemm_ln.hip
/opt/rocm/bin/hipcc $F -mllvm -amdgpu-mfma-vgpr-form=1 -DMIXDQ_STAMP=1 -c -o build/stamp/attention.o mixdq_amd/csrc/attention.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/stamp/libmixdq_stamp.so build/stamp/igemm.o build/stamp/igemm_ln.o \
  mixdq_amd/_obj/igemm_aq.o mixdq_amd/_obj/quantize.o mixdq_amd/_obj/iconv.o mixdq_amd/_obj/fused_norm.o build/stamp/attention.o
echo build/stamp/libmixdq_stamp.so
