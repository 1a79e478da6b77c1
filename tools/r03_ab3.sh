#!/bin/bash
# round-3 A/B run 3 (one box): attention variants + in-kernel stamps
R=$(pwd); out=$R/gpurun_out; log=$out/r03_ab3.log; : > $log
for rep in 1 2; do
  for lib in "" tools/ablate/librsvld_kpb1.so tools/ablate/librsvld_r02form.so; do
    echo "== attention, library: ${lib:-in-tree (bias step, no SLP, 8 waves from 16 384 tokens, 2 sub-tiles per barrier)} (pass $rep)" >> $log
    SHARED=1 HEADLINE=1 REPS=5 RSVLD_LIB=${lib:+$R/$lib} python3 tools/bench_attn.py >> $log 2>&1
  done
done
echo "== in-kernel stamps (diagnostic build)" >> $log
RSVLD_LIB=$R/tools/ablate/librsvld_stamp.so python3 tools/stamp_attn.py >> $log 2>&1
