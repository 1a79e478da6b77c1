#!/bin/bash
# round-3 run 11 (one box): ping-pong d = 64 kernel, one segment type at a time
R=$(pwd); out=$R/gpurun_out; log=$out/r03_ab11.log; : > $log
for lib in stamp8 stamp16; do
  echo "== stamps, library $lib" >> $log
  RSVLD_D64_KERNEL=c RSVLD_LIB=$R/tools/ablate/librsvld_$lib.so timeout -k 10 120 python3 tools/stamp_attn.py >> $log 2>&1
done
for lib in "" c_abl8 c_abl9 c_abl41 c_abl16; do
  echo "== d64c, library: ${lib:-in-tree}" >> $log
  RSVLD_D64_KERNEL=c ONLY64=1 HEADLINE=1 REPS=5 RSVLD_LIB=${lib:+$R/tools/ablate/librsvld_$lib.so} timeout -k 10 120 python3 tools/bench_attn.py >> $log 2>&1
done
