#!/bin/bash
# Diagnostic builds of the library with parts of attn_d512b_kernel / attn_d64b_kernel removed (results are WRONG by construction; timing only).
# usage: tools/ablate_attn.sh  -> tools/ablate/librsvld_<macro>_<N>.so (macro in lower case: a5b_abl / a6b_abl) for N in the list below
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SRC=$ROOT/remote-sensing-vision-language-diffusion-model_amd/csrc
mkdir -p $ROOT/tools/ablate
# MACRO=A5B_ABL (d = 512 kernel; list 1 4 5) or MACRO=A6B_ABL (d = 64 kernel; bits: 1 no exp, 2 no in-loop DMA / barrier, 8 no running
# max / row sum, 16 no V fragment reads, 32 no K fragment reads; lists used: "1 2 8 9 11" and "16 32 48 50 59")
MACRO=${MACRO:-A5B_ABL}
LIST=${LIST:-"1 4 5"}
TAG=$(echo $MACRO | tr A-Z a-z)
for N in $LIST; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -fno-gpu-rdc -fno-slp-vectorize -D$MACRO=$N -I$SRC -I$ROOT/include -c $SRC/attention.hip -o /tmp/attn_abl$N.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/tools/ablate/librsvld_${TAG}$N.so /tmp/attn_abl$N.o $SRC/build/conv_igemm.o $SRC/build/conv_halo.o $SRC/build/gemm.o $SRC/build/norm.o $SRC/build/elementwise.o $SRC/build/sampler.o $SRC/build/f32.o $SRC/build/gemv.o
done
ls -la $ROOT/tools/ablate
