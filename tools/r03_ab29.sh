#!/bin/bash
# round-3 run 29 (one box): loop-header alignment of the attention kernels (-mllvm -align-loops=64 / 256) vs the default
R=$(pwd); out=$R/gpurun_out; log=$out/r03_ab29.log; : > $log
for rep in 1 2; do
for lib in "" align64 align256; do
  echo "== library: ${lib:-in-tree} (pass $rep)" >> $log
  ONLY64=1 HEADLINE=1 REPS=5 RSVLD_LIB=${lib:+$R/tools/ablate/librsvld_$lib.so} timeout -k 10 120 python3 tools/bench_attn.py >> $log 2>&1
  SHARED=1 ONLY512=1 HEADLINE=1 REPS=3 RSVLD_LIB=${lib:+$R/tools/ablate/librsvld_$lib.so} timeout -k 10 200 python3 tools/bench_attn.py >> $log 2>&1
done
done
