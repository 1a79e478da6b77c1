#!/bin/bash
# round-3 A/B run 9 (one box): d = 64 attention, ping-pong kernel (attn_d64c) vs the four-waves-per-SIMD kernel (attn_d64b), same library
R=$(pwd); out=$R/gpurun_out; log=$out/r03_ab9.log; : > $log
timeout -k 10 300 python3 -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "attention_d64" >> $log 2>&1 || { echo "TESTS FAILED" >> $log; exit 1; }
for rep in 1 2; do
  for kern in b c; do
    echo "== d64 kernel $kern (pass $rep)" >> $log
    RSVLD_D64_KERNEL=$kern ONLY64=1 HEADLINE=1 REPS=5 timeout -k 10 120 python3 tools/bench_attn.py >> $log 2>&1
  done
done
RSVLD_D64_KERNEL=c ONLY64=1 REPS=5 timeout -k 10 120 python3 tools/bench_attn.py >> $log 2>&1
RSVLD_D64_KERNEL=b ONLY64=1 REPS=5 timeout -k 10 120 python3 tools/bench_attn.py >> $log 2>&1
