#!/bin/bash
# Round 6: phase stamps of the persistent four-phase kernel next to the one-tile kernel, and the f16in records
#   -> gpurun_out/r06_pp_stamps/
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
out=gpurun_out/r06_pp_stamps
rm -rf $out; mkdir -p $out
bash tools/stamp_build.sh > $out/build.log 2>&1 || tail -5 $out/build.log
export MIXDQ_HIP_LIB=$PWD/build/stamp/libmixdq_stamp.so
{
echo "== (8192, 10240, 1280) GEMM+GEGLU, one workgroup per tile (cfg 70)"; timeout 300 python tools/stamp_report.py 8192 10240 1280 --geglu --cfg 70 2>&1 | grep clock
echo "== the same launch, persistent (cfg 71): tile 0 = slots 1..7, tile 1 = slots 10, 11, 14"; timeout 300 python tools/stamp_report.py 8192 10240 1280 --geglu --cfg 71 --pp 2>&1 | grep clock
echo "== (8192, 3840, 1280) plain, cfg 70"; timeout 300 python tools/stamp_report.py 8192 3840 1280 --cfg 70 2>&1 | grep clock
echo "== persistent (cfg 71)"; timeout 300 python tools/stamp_report.py 8192 3840 1280 --cfg 71 --pp 2>&1 | grep clock
echo "== (32768, 5120, 640) GEMM+GEGLU cfg 70"; timeout 300 python tools/stamp_report.py 32768 5120 640 --geglu --cfg 70 2>&1 | grep clock
echo "== persistent (cfg 71)"; timeout 300 python tools/stamp_report.py 32768 5120 640 --geglu --cfg 71 --pp 2>&1 | grep clock
} > $out/stamps.txt
cat $out/stamps.txt
unset MIXDQ_HIP_LIB
for bs in 1 2 8; do
  echo "# tools/bench_f16in.py --bs $bs" >> $out/f16in.txt
  timeout 900 python tools/bench_f16in.py --bs $bs --L $([ $bs = 8 ] && echo 60 || echo 200) 2>/dev/null | grep "^{" >> $out/f16in.txt
done
echo "# tools/bench_f16in.py --chain (bs 1)" >> $out/f16in.txt
timeout 900 python tools/bench_f16in.py --chain 2>/dev/null | grep "^{" >> $out/f16in.txt
cat $out/f16in.txt | cut -c1-160
