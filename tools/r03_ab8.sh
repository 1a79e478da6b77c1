#!/bin/bash
# round-3 A/B run 8 (one box): gemm256 with asm LDS-DMA pieces (scalar base, branch-free steady state) vs the round-2 form (builtin)
R=$(pwd); out=$R/gpurun_out; log=$out/r03_ab8.log; : > $log
python3 -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize.py -m gpu -q -x >> $log 2>&1
for rep in 1 2; do
  for lib in "" tools/ablate/librsvld_gemm_builtin.so; do
    echo "== gemm256, library: ${lib:-in-tree (asm DMA pieces)} (pass $rep)" >> $log
    HEADLINE=1 REPS=5 RSVLD_LIB=${lib:+$R/$lib} python3 tools/bench_linear.py >> $log 2>&1
    echo "== same, with residual" >> $log
    RESIDUAL=1 HEADLINE=1 REPS=5 RSVLD_LIB=${lib:+$R/$lib} python3 tools/bench_linear.py >> $log 2>&1
  done
done
