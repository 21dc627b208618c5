#!/bin/bash
# Probes of single launches (parity subset, per-layer micro-benchmarks, in-kernel stamps):
#   bash tools/gpu_probe.sh <tag> [smoke] [stamps] [f16in] [ln]
tag=${1:?tag}; shift
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
for what in "$@"; do case $what in
  smoke) timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.txt 2>&1; tail -2 $out/smoke.txt ;;
  stamps)
    export MIXDQ_HIP_LIB=$PWD/build/stamp/libmixdq_stamp.so
    { for shape in "1 1024 1280" "1 4096 640" "8 1024 1280"; do
        echo "== attention $shape"; timeout 300 python tools/stamp_attn.py $shape 2>&1 | tail -3
        echo "== attention $shape --payload 24"; timeout 300 python tools/stamp_attn.py $shape --payload 24 2>&1 | tail -1
      done; } > $out/stamps_attn.txt 2>&1
    unset MIXDQ_HIP_LIB; cat $out/stamps_attn.txt ;;
  f16in) timeout 600 python tools/bench_f16in.py > $out/bench_f16in.txt 2>&1; grep '^{' $out/bench_f16in.txt ;;
  ln) timeout 600 python tools/bench_ln_gemm.py > $out/bench_ln_gemm.txt 2>&1; grep '^{' $out/bench_ln_gemm.txt ;;
esac; done
