#!/bin/bash
# round-3 run 19 (one box): gemm256 with ONE barrier per K tile (in-tree) vs two (twobar = the form shipped until now)
R=$(pwd); out=$R/gpurun_out; log=$out/r03_ab19.log; : > $log
timeout -k 10 600 python3 -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize.py -m gpu -q -x -k "gemm or linear or geglu or deterministic or identity" >> $log 2>&1 || { echo "TESTS FAILED" >> $log; tail -30 $log; exit 1; }
for rep in 1 2; do
  for lib in "" tools/ablate/librsvld_gemm_twobar.so; do
    echo "== gemm256, library: ${lib:-in-tree (one barrier per K tile)} (pass $rep)" >> $log
    HEADLINE=1 REPS=5 RSVLD_LIB=${lib:+$R/$lib} timeout -k 10 200 python3 tools/bench_linear.py >> $log 2>&1
  done
done
echo "== with residual, in-tree then twobar" >> $log
RESIDUAL=1 HEADLINE=1 REPS=5 timeout -k 10 200 python3 tools/bench_linear.py >> $log 2>&1
RESIDUAL=1 HEADLINE=1 REPS=5 RSVLD_LIB=$R/tools/ablate/librsvld_gemm_twobar.so timeout -k 10 200 python3 tools/bench_linear.py >> $log 2>&1
