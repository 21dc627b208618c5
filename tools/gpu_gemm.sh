#!/bin/bash
# operator parity (ops + fused) then the tile sweep of the UNet's GEMM / conv shapes
tag=${1:-gemm}
out=gpurun_out/r02_$tag
mkdir -p $out
( time timeout 1200 python -m pytest tests/test_ops_gpu.py tests/test_fused_gpu.py -x -q ) > $out/pytest_ops.log 2>&1
tail -6 $out/pytest_ops.log
timeout 900 python tools/bench_gemm.py --bs 1 > $out/gemm_lin_bs1.jsonl 2>&1
timeout 900 python tools/bench_gemm.py --bs 1 --conv > $out/gemm_conv_bs1.jsonl 2>&1
tail -3 $out/gemm_lin_bs1.jsonl | cut -c1-600
tail -3 $out/gemm_conv_bs1.jsonl | cut -c1-600
