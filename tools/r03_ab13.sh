#!/bin/bash
# round-3 run 13 (one box): d = 64 ping-pong kernel: priority and joint-softmax variants (in-tree: JOINT=1, PRIO=1)
R=$(pwd); out=$R/gpurun_out; log=$out/r03_ab13.log; : > $log
timeout -k 10 300 python3 -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "attention_d64" >> $log 2>&1 || { echo "TESTS FAILED" >> $log; exit 1; }
for rep in 1 2; do
for lib in "" c_prio0 c_prio2 c_joint0 c_joint0_prio0; do
  echo "== d64c, library: ${lib:-in-tree} (pass $rep)" >> $log
  RSVLD_D64_KERNEL=c ONLY64=1 HEADLINE=1 REPS=5 RSVLD_LIB=${lib:+$R/tools/ablate/librsvld_$lib.so} timeout -k 10 120 python3 tools/bench_attn.py >> $log 2>&1
done
done
echo "== d64b" >> $log
RSVLD_D64_KERNEL=b ONLY64=1 HEADLINE=1 REPS=5 timeout -k 10 120 python3 tools/bench_attn.py >> $log 2>&1
