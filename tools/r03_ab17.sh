#!/bin/bash
# round-3 run 17 (one box): d = 512 attention: softmax in 32 parts + permlane swap (in-tree), 8 parts + swap (coarse), 8 parts + ds_bpermute (shfl = the round-2 form)
R=$(pwd); out=$R/gpurun_out; log=$out/r03_ab17.log; : > $log
timeout -k 10 600 python3 -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize.py -m gpu -q -x -k "attention" >> $log 2>&1 || { echo "TESTS FAILED" >> $log; tail -30 $log; exit 1; }
for rep in 1 2; do
for lib in "" d512_coarse d512_shfl; do
  echo "== d512, library: ${lib:-in-tree (32 parts)} (pass $rep)" >> $log
  SHARED=1 ONLY512=1 HEADLINE=1 REPS=3 RSVLD_LIB=${lib:+$R/tools/ablate/librsvld_$lib.so} timeout -k 10 200 python3 tools/bench_attn.py >> $log 2>&1
done
done
