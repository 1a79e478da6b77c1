#!/bin/bash
# rocprofv3 evidence for the headline (c4) workload; run on the GPU box from the repo root:
#   bash tools/profile_c4.sh r04                         -> gpurun_out/r04_c4_{kernel_stats.csv,pmc_traffic.json}
#   bash tools/profile_c4.sh r04 split --precision split -> gpurun_out/r04_c4_split_{kernel_stats.csv,pmc_traffic.json}
# kernel stats of the bench command itself, then three separate --pmc passes (no tracing) over ONE iteration of each
# stage + the per-image fixed part (bench.py --pmc-pass).
set -e
tag=${1:-r04}
sfx=${2:+_$2}          # optional name suffix; everything behind it is passed to bench.py (e.g. --precision split)
shift; [ $# -gt 0 ] && shift
extra="$@"
tag=${tag}_c4${sfx}
R=$(pwd)
export TMPDIR=/tmp
out=$R/gpurun_out
rocprofv3 --kernel-trace --stats -d $out/${tag}_stats -o c4 --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras $extra > $out/${tag}_stats.log 2>&1
for c in FETCH_SIZE WRITE_SIZE "SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  n=$(echo $c | cut -d' ' -f1)
  # (--cached-cond: the counter passes serialise every dispatch, and the caption pass alone is ~200 k small PyTorch kernels;
  #  the kernels of THIS library run on the same shapes with or without it)
  rocprofv3 --pmc $c -d $out/${tag}_pmc_$n -o pmc --output-format csv -- python3 $R/bench.py --pmc-pass --cached-cond --no-cpu-baseline $extra > $out/${tag}_pmc_$n.log 2>&1
done
# the raw traces are far beyond gpurun's 64 MiB return limit: condense them here, keep only the summaries
python3 $R/tools/summarize_profiles.py --stats $out/${tag}_stats --out $out/${tag}_kernel_stats.csv
python3 $R/tools/summarize_profiles.py --pmc $out/${tag}_pmc_FETCH_SIZE $out/${tag}_pmc_WRITE_SIZE $out/${tag}_pmc_SQ_BUSY_CYCLES \
  --out $out/${tag}_pmc_traffic.json --command "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE | SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE (three separate passes, no tracing) --output-format csv -- python3 bench.py --pmc-pass --cached-cond --no-cpu-baseline $extra"
rm -rf $out/${tag}_stats $out/${tag}_pmc_FETCH_SIZE $out/${tag}_pmc_WRITE_SIZE $out/${tag}_pmc_SQ_BUSY_CYCLES
echo profile_c4 done
