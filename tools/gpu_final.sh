#!/bin/bash
# The round's record in one lease: counter passes of the dominant launches, the benchmark lines, the
# rocprofv3 kernel traces, then the whole GPU suite on the tree that produced them.
bash tools/pmc_r03.sh > gpurun_out/pmc_r03.log 2>&1
python3 tools/pmc_traffic_from_summary.py gpurun_out/r03_pmc/summary.json >> gpurun_out/pmc_r03.log 2>&1
cp profiles/pmc_traffic.json gpurun_out/r03_pmc/pmc_traffic.json
bash tools/gpu_round.sh r03_final3
tail -12 gpurun_out/pmc_r03.log
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
( time timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -4 ) > gpurun_out/r03_final3/pytest_gpu.txt 2>&1
cat gpurun_out/r03_final3/pytest_gpu.txt
