#!/bin/bash
# round-3 A/B run 1 (one box): d = 64 attention with / without the bias step, gemm256 with staggered first-round workgroups
R=$(pwd); out=$R/gpurun_out; log=$out/r03_ab1.log; : > $log
for rep in 1 2; do
  for lib in "" tools/ablate/librsvld_nobias.so; do
    echo "== attention d64, library: ${lib:-in-tree (A6B_BIAS=1)} (pass $rep)" >> $log
    HEADLINE=1 ONLY64=1 REPS=5 RSVLD_LIB=${lib:+$R/$lib} python3 tools/bench_attn.py >> $log 2>&1
  done
done
for lib in "" tools/ablate/librsvld_gstag6.so tools/ablate/librsvld_gstag13.so ""; do
  echo "== gemm256, library: ${lib:-in-tree}" >> $log
  HEADLINE=1 REPS=5 RSVLD_LIB=${lib:+$R/$lib} python3 tools/bench_linear.py >> $log 2>&1
done
