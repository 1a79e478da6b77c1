#!/bin/bash
# round-3 run 18 (one box): the cross-attention shapes of the headline (77 keys) on attn_d64b vs attn_d64c
R=$(pwd); out=$R/gpurun_out; log=$out/r03_ab18.log; : > $log
for rep in 1 2; do
  for kern in b c; do
    echo "== d64 kernel $kern, 77 keys (pass $rep)" >> $log
    CROSS=77 RSVLD_D64_KERNEL=$kern ONLY64=1 HEADLINE=1 REPS=20 timeout -k 10 120 python3 tools/bench_attn.py >> $log 2>&1
  done
done
