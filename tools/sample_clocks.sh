#!/bin/bash
# Board power and sclk (rocm-smi, 1 s samples) while one attention micro-benchmark loops: shows whether a kernel runs at the
# board's power limit (the chip then trades clock for activity: MI355X_MICROARCH.md "DVFS give-back").
# usage: tools/sample_clocks.sh d64|d512     (writes gpurun_out/clk_<which>.txt)
W=${1:-d64}
mkdir -p gpurun_out
if [ $W = d64 ]; then export ONLY64=1 REPS=900; else export ONLY512=1 SHARED=1 REPS=120; fi
( HEADLINE=1 timeout -k 10 200 python tools/bench_attn.py > gpurun_out/clk_bench_$W.txt 2>&1 ) &
BP=$!
for i in $(seq 1 45); do
  kill -0 $BP 2>/dev/null || break
  rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|Power (W)" | sed 's/.*: //' | tr '\n' ' '; echo
  sleep 1
done > gpurun_out/clk_$W.txt 2>&1
wait $BP
grep "TF/s" gpurun_out/clk_bench_$W.txt
sort -t'(' -k2 -n gpurun_out/clk_$W.txt | awk '{print}' | sort -k3 -n | tail -8
