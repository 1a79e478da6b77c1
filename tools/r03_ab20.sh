#!/bin/bash
# round-3 run 20 (one box): gemm256 one barrier per K tile (in-tree) vs two (twobar), 20 repetitions per shape, A-B-A-B
R=$(pwd); out=$R/gpurun_out; log=$out/r03_ab20.log; : > $log
for rep in 1 2 3; do
  for lib in "" tools/ablate/librsvld_gemm_twobar.so; do
    echo "== gemm256, library: ${lib:-in-tree (one barrier per K tile)} (pass $rep)" >> $log
    HEADLINE=1 REPS=20 RSVLD_LIB=${lib:+$R/$lib} timeout -k 10 200 python3 tools/bench_linear.py >> $log 2>&1
  done
done
