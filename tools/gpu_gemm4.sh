#!/bin/bash
out=gpurun_out/r02_gemm4
mkdir -p $out
CF=4,37,41,35,25,42,45,44,56,6,3,13
timeout 900 python tools/bench_gemm.py --bs 1 --w4 --cfgs $CF > $out/gemm_lin_bs1_w4.jsonl 2>&1
timeout 900 python tools/bench_gemm.py --bs 1 --w4 --conv --cfgs $CF > $out/gemm_conv_bs1_w4.jsonl 2>&1
