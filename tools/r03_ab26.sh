#!/bin/bash
# round-3 run 26 (one box): d = 64 kernels, O staged through LDS to whole-row stores (in-tree) vs 8-byte row-strided stores (nostage)
R=$(pwd); out=$R/gpurun_out; log=$out/r03_ab26.log; : > $log
timeout -k 10 300 python3 -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "attention_d64" >> $log 2>&1 || { echo "TESTS FAILED" >> $log; tail -30 $log; exit 1; }
for rep in 1 2; do
for lib in "" nostage; do
  echo "== library: ${lib:-in-tree (staged O)} (pass $rep): cross-attention (77 keys, attn_d64b), then self-attention (attn_d64c)" >> $log
  CROSS=77 ONLY64=1 HEADLINE=1 REPS=20 RSVLD_LIB=${lib:+$R/tools/ablate/librsvld_$lib.so} timeout -k 10 120 python3 tools/bench_attn.py >> $log 2>&1
  ONLY64=1 HEADLINE=1 REPS=5 RSVLD_LIB=${lib:+$R/tools/ablate/librsvld_$lib.so} timeout -k 10 120 python3 tools/bench_attn.py >> $log 2>&1
done
done
