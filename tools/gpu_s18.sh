#!/bin/bash
# session 18: board power and clocks while the batch-8 / batch-1 step replays (is the step power-limited?)
out=gpurun_out/s18
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
rocm-smi --showpower --showclocks --showmaxpower > $out/idle.txt 2>&1
for bs in 8 1; do
  steps=600; [ $bs = 1 ] && steps=2500
  timeout 600 python bench.py --no-fp16 --no-cpu-baseline --no-roofline --no-batch8 --no-dropin --steps $steps --batch $bs > $out/bench_bs$bs.json 2>/dev/null &
  pid=$!
  sleep 45
  for i in 1 2 3 4 5 6 7 8; do
    rocm-smi --showpower --showclocks 2>&1 | grep -i -E "power|sclk|mclk|fclk" | tr '\n' '|' >> $out/load_bs$bs.txt
    echo >> $out/load_bs$bs.txt
    sleep 2
  done
  wait $pid
done
cat $out/idle.txt | head -30; echo; cat $out/load_bs8.txt; echo; cat $out/load_bs1.txt; tail -c 300 $out/bench_bs8.json
