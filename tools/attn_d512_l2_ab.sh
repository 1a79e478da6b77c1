#!/bin/bash
# tools/attn_d512_l2_ab.sh [passes]: what does the L2-miss traffic of attn_d512b cost?  A-B-A on one box: the shipped kernel against the
# diagnostic build whose key stream cycles over 2 MiB (A5B_KEYWRAP = 64: L2-resident in every XCD; same loop, same LDS traffic, same
# MFMAs, results wrong), at the two Stage-1 shapes of the headline (65 536 and 262 144 keys, keys = values = one tensor), plus the PMC
# FETCH_SIZE of both builds.  Run from the repo root on the GPU box; the variant library is built beforehand (no GPU needed):
#   tools/build_attn_variant.sh keywrap -DA5B_KEYWRAP=64
set -e
n=${1:-2}
out=gpurun_out/r05_attn_d512_l2_resident_ab.txt
: > $out
for pass in $(seq 1 $n); do
  for v in shipped keywrap; do
    echo "== pass $pass, $v" >> $out
    if [ $v = keywrap ]; then export RSVLD_LIB=$PWD/tools/ablate/librsvld_keywrap.so; else unset RSVLD_LIB; fi
    HEADLINE=1 ONLY512=1 SHARED=1 REPS=${REPS:-5} timeout -k 10 300 python3 tools/bench_attn.py >> $out 2>&1 || { echo "BENCH FAILED" >> $out; tail -5 $out; exit 1; }
  done
done
unset RSVLD_LIB
cat $out
