#!/bin/bash
# round-3 A/B run 5 (one box): d = 64 request phase with / without the per-tile 64-bit multiplies
R=$(pwd); out=$R/gpurun_out; log=$out/r03_ab5.log; : > $log
for rep in 1 2; do
  for lib in "" tools/ablate/librsvld_mulptr.so; do
    echo "== attention d64, library: ${lib:-in-tree (pointer increments)} (pass $rep)" >> $log
    HEADLINE=1 ONLY64=1 REPS=5 RSVLD_LIB=${lib:+$R/$lib} python3 tools/bench_attn.py >> $log 2>&1
  done
done
