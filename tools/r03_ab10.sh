#!/bin/bash
# round-3 run 10 (one box): where the ping-pong d = 64 kernel spends its segments (stamps) + ablations
R=$(pwd); out=$R/gpurun_out; log=$out/r03_ab10.log; : > $log
RSVLD_D64_KERNEL=c RSVLD_LIB=$R/tools/ablate/librsvld_stamp.so timeout -k 10 120 python3 tools/stamp_attn.py >> $log 2>&1
for lib in "" c_abl1 c_abl2 c_abl6 c_abl7; do
  echo "== d64c, library: ${lib:-in-tree}" >> $log
  RSVLD_D64_KERNEL=c ONLY64=1 HEADLINE=1 REPS=5 RSVLD_LIB=${lib:+$R/tools/ablate/librsvld_$lib.so} timeout -k 10 120 python3 tools/bench_attn.py >> $log 2>&1
done
