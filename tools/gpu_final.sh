#!/bin/bash
# The round's record in one lease: counter passes of the dominant launches, then the benchmark lines.
bash tools/pmc_r03.sh > gpurun_out/pmc_r03.log 2>&1
python3 tools/pmc_traffic_from_summary.py gpurun_out/r03_pmc/summary.json >> gpurun_out/pmc_r03.log 2>&1
cp profiles/pmc_traffic.json gpurun_out/r03_pmc/pmc_traffic.json
bash tools/gpu_round.sh r03_final2
tail -12 gpurun_out/pmc_r03.log
