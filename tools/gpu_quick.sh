#!/bin/bash
# quick GPU check of a change: the named test files, then the quantized bench leg only
timeout 1500 python -m pytest "$@" -x -q -m gpu 2>&1 | tail -6
timeout 600 python bench.py --no-fp16 --no-cpu-baseline --no-roofline --steps 30 2>/dev/null | python3 -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r['ms_per_step'], r['config'])"
