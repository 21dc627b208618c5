#!/bin/bash
tag=${1:-gemm3}
out=gpurun_out/r02_$tag
mkdir -p $out
CF=13,14,15,18,20,25,35,41,44,46,53,54,55,7,3
timeout 1200 python tools/bench_gemm.py --bs 8 --cfgs $CF > $out/gemm_lin_bs8.jsonl 2>&1
timeout 1200 python tools/bench_gemm.py --bs 8 --conv --cfgs $CF > $out/gemm_conv_bs8.jsonl 2>&1
tail -1 $out/gemm_lin_bs8.jsonl; tail -1 $out/gemm_conv_bs8.jsonl
