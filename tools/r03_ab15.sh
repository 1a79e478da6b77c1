#!/bin/bash
# round-3 run 15 (one box): pipelined kernel ablations (1 no in-loop DMA, 2 no softmax chunks)
R=$(pwd); out=$R/gpurun_out; log=$out/r03_ab15.log; : > $log
for lib in "" p_abl1 p_abl2 p_abl3; do
  echo "== d64p, library: ${lib:-in-tree}" >> $log
  RSVLD_D64_KERNEL=p ONLY64=1 HEADLINE=1 REPS=5 RSVLD_LIB=${lib:+$R/tools/ablate/librsvld_$lib.so} timeout -k 10 120 python3 tools/bench_attn.py >> $log 2>&1
done
