#!/bin/bash
# Round 6: gn_finalize_kernel with every partial of a lane requested before the first is used (one trip to the memory
# side instead of eight): the GroupNorm tests, then a same-box A/B of the two libraries.  -> gpurun_out/r06_gnfin/
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
mkdir -p gpurun_out/r06_gnfin
timeout 600 python -m pytest tests/test_fused_gpu.py tests/test_glue_gpu.py tests/test_ops_gpu.py -q -m gpu -k "groupnorm or GroupNorm or gn_ or norm" 2>&1 | tail -3 | tee gpurun_out/r06_gnfin/pytest.txt
sed -i 's/for rep in 1 2 3; do/for rep in 1 2; do/' tools/gpu_ab_lib.sh
bash tools/gpu_ab_lib.sh r06_gnfin --no-batch8
