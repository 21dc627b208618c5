#!/bin/bash
# session 3: new parity tests, the new bench line, 256x160 tile, tile-map super-row sweep, stamps, PMC r04, full suite
out=gpurun_out/s3
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
( time timeout 1200 python -m pytest tests/test_large_gpu.py tests/test_unet_path_a_gpu.py -q -m gpu -x -s -k "shard_size or over_4_gib or path_a" 2>&1 | tail -15 ) > $out/pytest_new.txt 2>&1
( timeout 600 python -m pytest tests/test_fused_gpu.py tests/test_ops_gpu.py tests/test_cabi.py -q -m gpu -x 2>&1 | tail -5 ) > $out/pytest_ops.txt 2>&1
( time timeout 1500 python bench.py > $out/bench_default.json 2> $out/bench_default.err ) > $out/bench_default.time 2>&1
timeout 600 python tools/bench_geglu_cfgs.py > $out/geglu_cfgs.txt 2>&1
timeout 300 python tools/bench_geglu_cfgs.py --bs8 >> $out/geglu_cfgs.txt 2>&1
for gm in 8 4 16 2; do
  echo "== MIXDQ_IGEMM_GM=$gm" >> $out/gm_sweep.txt
  MIXDQ_IGEMM_GM=$gm timeout 300 python tools/bench_gemm.py --bs 8 --cfgs 70,25 2>&1 | grep -v amdgpu | cut -c1-200 >> $out/gm_sweep.txt
done
export MIXDQ_HIP_LIB=$PWD/build/stamp/libmixdq_stamp.so
for spec in "1024 10240 1280 --geglu --cfg 25 --cold" "1024 10240 1280 --geglu --cfg 26 --cold" "8192 10240 1280 --geglu --cfg 70" "8192 3840 1280 --cfg 70" "1024 1280 1280 --cfg 56 --res --cold" "1024 1280 5120 --cfg 45 --res --cold" "8192 1280 1280 --cfg 25 --res"; do
  echo "== $spec" >> $out/stamps.txt
  timeout 300 python tools/stamp_report.py $spec 2>&1 | grep -v amdgpu.ids | tail -2 >> $out/stamps.txt
done
unset MIXDQ_HIP_LIB
timeout 1500 bash tools/pmc_r04.sh > $out/pmc.txt 2>&1
( time timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -6 ) > $out/pytest_all.txt 2>&1
cat $out/pytest_new.txt $out/pytest_ops.txt $out/geglu_cfgs.txt $out/stamps.txt; tail -20 $out/pmc.txt; cat $out/pytest_all.txt
