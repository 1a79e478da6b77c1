#!/bin/bash
# round-3 run 24 (one box): ping-pong kernel with a branch-free steady-state copy of its loop (in-tree) vs the general loop only (c_nosteady)
R=$(pwd); out=$R/gpurun_out; log=$out/r03_ab24.log; : > $log
timeout -k 10 300 python3 -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "attention_d64" >> $log 2>&1 || { echo "TESTS FAILED" >> $log; tail -30 $log; exit 1; }
for rep in 1 2 3; do
for lib in "" c_nosteady; do
  echo "== d64c, library: ${lib:-in-tree (steady-state loop)} (pass $rep)" >> $log
  RSVLD_D64_KERNEL=c ONLY64=1 HEADLINE=1 REPS=5 RSVLD_LIB=${lib:+$R/tools/ablate/librsvld_$lib.so} timeout -k 10 120 python3 tools/bench_attn.py >> $log 2>&1
done
done
