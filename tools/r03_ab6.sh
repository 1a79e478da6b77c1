#!/bin/bash
# round-3 A/B run 6 (one box): d = 64 attention with / without the sum-checked softmax
R=$(pwd); out=$R/gpurun_out; log=$out/r03_ab6.log; : > $log
for rep in 1 2; do
  for lib in "" tools/ablate/librsvld_nosumchk.so; do
    echo "== attention d64, library: ${lib:-in-tree (sum-checked softmax)} (pass $rep)" >> $log
    HEADLINE=1 ONLY64=1 REPS=5 RSVLD_LIB=${lib:+$R/$lib} python3 tools/bench_attn.py >> $log 2>&1
  done
done
