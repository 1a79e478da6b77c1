#!/bin/bash
# round-3 run 30 (one box): XCD-aware work order in the ping-pong kernel (in-tree) vs the plain grid order (c_noxcd)
R=$(pwd); out=$R/gpurun_out; log=$out/r03_ab30.log; : > $log
timeout -k 10 300 python3 -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "attention_d64" >> $log 2>&1 || { echo "TESTS FAILED" >> $log; tail -30 $log; exit 1; }
for rep in 1 2; do
for lib in "" c_noxcd; do
  echo "== library: ${lib:-in-tree (XCD order)} (pass $rep)" >> $log
  RSVLD_D64_KERNEL=c RSVLD_LIB=${lib:+$R/tools/ablate/librsvld_$lib.so} timeout -k 10 120 python3 tools/bench_attn_scan.py >> $log 2>&1
done
done
