#!/bin/bash
# round-3 run 27 (one box): is the current attention build slower than the build of run 24 (commit 9c9bd4e) on the self-attention shapes?
R=$(pwd); out=$R/gpurun_out; log=$out/r03_ab27.log; : > $log
for rep in 1 2 3; do
for lib in "" attn_run24 nostage; do
  echo "== library: ${lib:-in-tree} (pass $rep)" >> $log
  ONLY64=1 HEADLINE=1 REPS=5 RSVLD_LIB=${lib:+$R/tools/ablate/librsvld_$lib.so} timeout -k 10 120 python3 tools/bench_attn.py >> $log 2>&1
done
done
