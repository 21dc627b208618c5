#!/bin/bash
run() { echo "== $1"; MIXDQ_IGEMM_TUNE="$1" timeout 900 python bench.py --batch 8 --no-fp16 --no-cpu-baseline --no-roofline --steps 10 2>/dev/null | python3 -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r['ms_per_step'])"; }
run "8192x10240x1280=13,32768x5120x640=13"
run "8192x3840x1280=13,32768x1920x640=13"
run "8192x10240x1280=13,8192x3840x1280=13,32768x1920x640=13,32768x5120x640=13"
run ""
