#!/bin/bash
# Round-5 experiments on attn_d64c, A-B-A on ONE box: every variant first passes the bit-identity test against
# attn_d64b (tests/test_gpu_kernels.py -k pingpong) with ITS library, then is timed at the two headline shapes.
#   tools/attn_d64c_ab.sh VARIANT...   (tools/ablate/librsvld_<VARIANT>.so, built by tools/build_attn_variant.sh)
set -u
R=$(cd "$(dirname "$0")/.." && pwd); cd "$R"
mkdir -p gpurun_out; log=gpurun_out/r05_attn_d64_ab.txt; : > "$log"
for v in "$@"; do
  lib=$R/tools/ablate/librsvld_$v.so
  echo "== parity: $v" >> "$log"
  RSVLD_LIB=$lib timeout -k 10 300 python3 -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "pingpong or attention_d64" 2>&1 | tail -3 >> "$log" || { echo "PARITY FAILED ($v)" >> "$log"; }
done
for rep in 1 2 3; do
  for v in "$@"; do
    echo "== variant: $v (pass $rep)" >> "$log"
    HEADLINE=1 ONLY64=1 REPS=20 RSVLD_LIB=$R/tools/ablate/librsvld_$v.so timeout -k 10 300 python3 tools/bench_attn.py >> "$log" 2>&1 || { echo "BENCH FAILED ($v)" >> "$log"; exit 1; }
  done
done
for v in d64c_stamp d64c_vinm_stamp; do
  [ -f tools/ablate/librsvld_$v.so ] || continue
  echo "== stamps: $v" >> "$log"
  RSVLD_D64_KERNEL=c RSVLD_LIB=$R/tools/ablate/librsvld_$v.so timeout -k 10 300 python3 tools/stamp_attn.py >> "$log" 2>&1 || echo "STAMPS FAILED" >> "$log"
done
cat "$log"
