#!/bin/bash
# round-3 run 16 (one box): ping-pong kernel knobs: LDS-DMA request in the vector segment, M0 declared clobbered
R=$(pwd); out=$R/gpurun_out; log=$out/r03_ab16.log; : > $log
for rep in 1 2; do
for lib in "" c_dmav c_m0 c_m0_dmav; do
  echo "== d64c, library: ${lib:-in-tree} (pass $rep)" >> $log
  RSVLD_D64_KERNEL=c ONLY64=1 HEADLINE=1 REPS=5 RSVLD_LIB=${lib:+$R/tools/ablate/librsvld_$lib.so} timeout -k 10 120 python3 tools/bench_attn.py >> $log 2>&1
done
done
for lib in c_dmav c_m0_dmav; do
  echo "== tests with $lib" >> $log
  RSVLD_LIB=$R/tools/ablate/librsvld_$lib.so timeout -k 10 300 python3 -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "attention_d64" >> $log 2>&1
done
