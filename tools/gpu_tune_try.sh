#!/bin/bash
# A/B tile overrides for single shapes inside the whole captured step (MIXDQ_IGEMM_TUNE)
run() { echo "== $1"; MIXDQ_IGEMM_TUNE="$1" timeout 600 python bench.py --no-fp16 --no-cpu-baseline --no-roofline --steps 30 2>/dev/null | python3 -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r['ms_per_step'])"; }
for c in "$@"; do run "$c"; done
