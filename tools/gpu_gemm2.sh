#!/bin/bash
tag=${1:-gemm2}
out=gpurun_out/r02_$tag
mkdir -p $out
CF=4,37,41,35,25,42,45,43,44,46,49,50,51,52,53,54,55,56
timeout 900 python tools/bench_gemm.py --bs 1 --cfgs $CF > $out/gemm_lin_bs1.jsonl 2>&1
timeout 900 python tools/bench_gemm.py --bs 1 --conv --cfgs $CF > $out/gemm_conv_bs1.jsonl 2>&1
for c in 41 42 49 50 56 4 51; do timeout 120 python tools/bench_cold.py $c; done > $out/cold.txt 2>&1
cat $out/cold.txt
