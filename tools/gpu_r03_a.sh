#!/bin/bash
# Round 3, lease A: full GPU suite (no -x: see every failure), chain / cold-weight / prefetch
# microbenchmark, attention kernel choice at batch 1 / 2 / 8, default bench line.
out=gpurun_out/r03_a
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
( time timeout 2400 python -m pytest tests -m gpu -q ) > $out/pytest.log 2>&1
tail -25 $out/pytest.log
timeout 900 python tools/bench_chain.py > $out/chain.log 2>&1
cat $out/chain.log
for bs in 1 2 8; do for cfg in 0 8 4; do
  echo "== attn bs $bs cfg $cfg" >> $out/attn.log
  timeout 300 python tools/bench_attn.py --bs $bs --impl hip --cfg $cfg >> $out/attn.log 2>&1
done; done
cat $out/attn.log
timeout 900 python bench.py --no-fp16 --no-cpu-baseline > $out/bench_bs1.json 2> $out/bench_bs1.err
tail -c 2500 $out/bench_bs1.json
