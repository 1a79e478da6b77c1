#!/bin/bash
# round-3 A/B run 4 (one box): 16-wave d = 64 attention; kernel statistics of the caption pass
R=$(pwd); out=$R/gpurun_out; log=$out/r03_ab4.log; : > $log
export TMPDIR=/tmp
for rep in 1 2; do
  for lib in "" tools/ablate/librsvld_nw16.so; do
    echo "== attention d64, library: ${lib:-in-tree} (pass $rep)" >> $log
    HEADLINE=1 ONLY64=1 REPS=5 RSVLD_LIB=${lib:+$R/$lib} python3 tools/bench_attn.py >> $log 2>&1
  done
done
RSVLD_LIB=$R/tools/ablate/librsvld_nw16.so python3 -m pytest tests/test_gpu_kernels.py -m gpu -q -k "attention_d64" >> $log 2>&1
rocprofv3 --kernel-trace --stats -d $out/r03_caption_stats -o cap --output-format csv -- python3 $R/tools/profile_caption.py > $out/r03_caption_stats.log 2>&1
python3 $R/tools/summarize_profiles.py --stats $out/r03_caption_stats --out $out/r03_caption_kernel_stats.csv >> $log 2>&1
rm -rf $out/r03_caption_stats
