#!/bin/bash
# round-3 run 12 (one box): d = 64 ping-pong kernel, one-barrier form: tests, A/B against attn_d64b, stamps, ablations
R=$(pwd); out=$R/gpurun_out; log=$out/r03_ab12.log; : > $log
timeout -k 10 300 python3 -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "attention_d64" >> $log 2>&1 || { echo "TESTS FAILED" >> $log; exit 1; }
for rep in 1 2; do
  for kern in b c; do
    echo "== d64 kernel $kern (pass $rep)" >> $log
    RSVLD_D64_KERNEL=$kern ONLY64=1 HEADLINE=1 REPS=5 timeout -k 10 120 python3 tools/bench_attn.py >> $log 2>&1
  done
done
RSVLD_D64_KERNEL=c RSVLD_LIB=$R/tools/ablate/librsvld_stamp.so timeout -k 10 120 python3 tools/stamp_attn.py >> $log 2>&1
for lib in c_abl1 c_abl8 c_abl9; do
  echo "== d64c, library: ${lib:-in-tree}" >> $log
  RSVLD_D64_KERNEL=c ONLY64=1 HEADLINE=1 REPS=5 RSVLD_LIB=${lib:+$R/tools/ablate/librsvld_$lib.so} timeout -k 10 120 python3 tools/bench_attn.py >> $log 2>&1
done
