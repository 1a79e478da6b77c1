#!/bin/bash
# round-3 A/B run 7 (one box): LDS-DMA statements with M0 declared clobbered (tools/audit_m0.py green) vs saved / restored
R=$(pwd); out=$R/gpurun_out; log=$out/r03_ab7.log; : > $log
for rep in 1 2; do
  for lib in "" tools/ablate/librsvld_m0clobber.so; do
    echo "== attention, library: ${lib:-in-tree (M0 saved / restored around every piece)} (pass $rep)" >> $log
    SHARED=1 HEADLINE=1 REPS=5 RSVLD_LIB=${lib:+$R/$lib} python3 tools/bench_attn.py >> $log 2>&1
  done
done
RSVLD_LIB=$R/tools/ablate/librsvld_m0clobber.so python3 -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize.py -m gpu -q -k "attention or deterministic" >> $log 2>&1
