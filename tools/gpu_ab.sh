#!/bin/bash
# A/B of tile choices inside the whole UNet (bench.py, W8A8 leg only)
out=gpurun_out/r02_ab
mkdir -p $out
run() { tag=$1; shift; env "$@" timeout 600 python bench.py --no-fp16 --no-cpu-baseline --no-roofline --steps 30 > $out/$tag.json 2> $out/$tag.err; python3 -c "import json;d=json.load(open('$out/$tag.json'));print('$tag', round(d['ms_per_step'],3))"; }
run new A=1
run old MIXDQ_IGEMM_TUNE=1024x1280x1280=41,1024x1280x5120=41,1024x1280x11520=37,1024x1280x23040=37,4096x640x2560=35,4096x640x5760=35,4096x1280x11520=35,16384x320x5760=35,16384x320x2880=35,16384x320x8640=35,4096x640x17280=35
run c42 MIXDQ_IGEMM_TUNE=1024x1280x1280=42
run c41 MIXDQ_IGEMM_TUNE=1024x1280x1280=41
run c49 MIXDQ_IGEMM_TUNE=1024x1280x1280=49
run new2 A=1
