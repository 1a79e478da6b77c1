#!/bin/bash
# round-3 run 28 (one box): redo / rescale branches marked cold (in-tree) vs the build of run 24 (commit 9c9bd4e)
R=$(pwd); out=$R/gpurun_out; log=$out/r03_ab28.log; : > $log
timeout -k 10 600 python3 -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize.py -m gpu -q -x -k "attention" >> $log 2>&1 || { echo "TESTS FAILED" >> $log; tail -30 $log; exit 1; }
for rep in 1 2; do
for lib in "" attn_run24; do
  echo "== library: ${lib:-in-tree} (pass $rep)" >> $log
  ONLY64=1 HEADLINE=1 REPS=5 RSVLD_LIB=${lib:+$R/tools/ablate/librsvld_$lib.so} timeout -k 10 120 python3 tools/bench_attn.py >> $log 2>&1
  SHARED=1 ONLY512=1 HEADLINE=1 REPS=3 RSVLD_LIB=${lib:+$R/tools/ablate/librsvld_$lib.so} timeout -k 10 200 python3 tools/bench_attn.py >> $log 2>&1
done
done
