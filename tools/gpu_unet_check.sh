cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
mkdir -p gpurun_out/chk
timeout 900 python -m pytest tests/test_unet_gpu.py tests/test_unet_full_gpu.py -q -m gpu -x 2>&1 | tail -3
timeout 600 python bench.py --no-fp16 --no-cpu-baseline --no-roofline > gpurun_out/chk/bs1.json 2> gpurun_out/chk/err.txt
python -c "
import json; d=json.loads(open('gpurun_out/chk/bs1.json').read().strip().splitlines()[-1]); print('bs1 ms', d['ms_per_step'])"
