#!/bin/bash
# Evidence for the two attention kernels at the HEADLINE shapes (run on the GPU box from the repo root):
#   bash tools/profile_attn.sh r02
# 1. throughput of the shipped library and of the ablation builds (tools/ablate_attn.sh; built beforehand, they travel
#    with the snapshot) in ONE process sequence on ONE box;  2. rocprofv3 PMC passes (no tracing): wave-cycle buckets, LDS
#    instruction / bank-conflict / array-busy counters, MFMA-busy.  Raw counter files are condensed here.
set -e
tag=${1:-r02}
R=$(pwd)
export TMPDIR=/tmp
out=$R/gpurun_out
log=$out/${tag}_attn_ablation.log
: > $log
if ! ls tools/ablate/librsvld_a5b_abl*.so tools/ablate/librsvld_a6b_abl*.so > /dev/null 2>&1; then
  echo "no ablation library under tools/ablate/ (build them first: MACRO=A5B_ABL LIST='1 4 5' tools/ablate_attn.sh; MACRO=A6B_ABL LIST='1 2 8 11' tools/ablate_attn.sh)" >&2
  exit 1
fi
echo "== shipped library" >> $log
HEADLINE=1 python3 tools/bench_attn.py >> $log 2>&1
for lib in $(ls tools/ablate/librsvld_a5b_abl*.so 2>/dev/null); do
  echo "== $lib (d = 512 kernel ablated: 1 no softmax VALU, 4 no in-loop DMA, 5 both)" >> $log
  HEADLINE=1 ONLY512=1 RSVLD_LIB=$R/$lib python3 tools/bench_attn.py >> $log 2>&1
done
for lib in $(ls tools/ablate/librsvld_a6b_abl*.so 2>/dev/null); do
  echo "== $lib (d = 64 kernel ablated: 1 no exp, 2 no in-loop DMA / barrier, 8 no max / row sum, 11 all)" >> $log
  HEADLINE=1 ONLY64=1 RSVLD_LIB=$R/$lib python3 tools/bench_attn.py >> $log 2>&1
done
echo "== shipped library again (drift check)" >> $log
HEADLINE=1 python3 tools/bench_attn.py >> $log 2>&1
rocprofv3 -L > $out/${tag}_counters_list.txt 2>&1 || true
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE"
P2="SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE"
i=0
dirs=""
for c in "$P1" "$P2"; do
  i=$((i+1))
  if HEADLINE=1 REPS=1 rocprofv3 --pmc $c -d $out/${tag}_attn_pmc_$i -o pmc --output-format csv -- python3 $R/tools/bench_attn.py > $out/${tag}_attn_pmc_$i.log 2>&1; then
    dirs="$dirs $out/${tag}_attn_pmc_$i"
  else
    echo "pmc pass $i failed (see its log)"; tail -3 $out/${tag}_attn_pmc_$i.log
  fi
done
if [ -n "$dirs" ]; then
  python3 $R/tools/summarize_profiles.py --pmc $dirs --out $out/${tag}_attn_pmc.json --command "HEADLINE=1 REPS=1 rocprofv3 --pmc {$P1 | $P2} (separate passes, no tracing) -- python3 tools/bench_attn.py"
fi
rm -rf $out/${tag}_attn_pmc_1 $out/${tag}_attn_pmc_2
grep -c . $out/${tag}_counters_list.txt > /dev/null && head -c 20000 $out/${tag}_counters_list.txt > $out/${tag}_counters_list_head.txt && rm $out/${tag}_counters_list.txt
echo profile_attn done
