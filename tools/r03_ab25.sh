#!/bin/bash
# round-3 run 25 (one box): d = 512 ablations of the final form (1 no softmax VALU, 4 no in-loop DMA, 16 no per-tile barrier; results wrong, timing only)
R=$(pwd); out=$R/gpurun_out; log=$out/r03_ab25.log; : > $log
for lib in "" d512_abl1 d512_abl4 d512_abl16; do
  echo "== d512, library: ${lib:-in-tree}" >> $log
  SHARED=1 ONLY512=1 HEADLINE=1 REPS=3 RSVLD_LIB=${lib:+$R/tools/ablate/librsvld_$lib.so} timeout -k 10 200 python3 tools/bench_attn.py >> $log 2>&1
done
