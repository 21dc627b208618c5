#!/bin/bash
# The benchmark lines and kernel traces (tools/gpu_round.sh) and the whole GPU suite, without the counter passes.
bash tools/gpu_round.sh r03_final4
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
( time timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -4 ) > gpurun_out/r03_final4/pytest_gpu.txt 2>&1
cat gpurun_out/r03_final4/pytest_gpu.txt
