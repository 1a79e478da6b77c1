#!/bin/bash
# tools/gemm_ab.sh [passes]: gemm256 persistent form (the library's choice) against the one-tile form (RSVLD_GEMM256_ONE_TILE=1) on the nine
# headline shapes, with and without residual, A-B-A-B on one box; torch.matmul beside both as the calibration.  Log: gpurun_out/gemm_ab.log
set -u
R=$(cd "$(dirname "$0")/.." && pwd); cd "$R"
passes=${1:-2}; log=gpurun_out/gemm_ab.log; mkdir -p gpurun_out; : > "$log"
for rep in $(seq 1 "$passes"); do
  for v in persist one_tile; do
    for r in "" 1; do
      echo "== $v ${r:+residual }(pass $rep)" >> "$log"
      env HEADLINE=1 REPS=${REPS:-10} ${r:+RESIDUAL=1} $([ $v = one_tile ] && echo RSVLD_GEMM256_ONE_TILE=1) timeout -k 10 300 python3 tools/bench_linear.py >> "$log" 2>&1 || { echo "BENCH FAILED" >> "$log"; tail -20 "$log"; exit 1; }
    done
  done
done
