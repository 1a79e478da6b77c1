#!/bin/bash
# round-3 run 14 (one box): d = 64 pipelined kernel (attn_d64p): tests, A/B against b and c
R=$(pwd); out=$R/gpurun_out; log=$out/r03_ab14.log; : > $log
timeout -k 10 300 python3 -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "attention_d64" >> $log 2>&1 || { echo "TESTS FAILED" >> $log; tail -30 $log; exit 1; }
for rep in 1 2; do
  for kern in b c p; do
    echo "== d64 kernel $kern (pass $rep)" >> $log
    RSVLD_D64_KERNEL=$kern ONLY64=1 HEADLINE=1 REPS=5 timeout -k 10 120 python3 tools/bench_attn.py >> $log 2>&1
  done
done
