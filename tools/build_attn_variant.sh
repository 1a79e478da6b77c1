#!/bin/bash
# tools/build_attn_variant.sh <name> <hipcc flags...>  ->  tools/ablate/librsvld_<name>.so  (attention.hip rebuilt with the flags, the other objects of the in-tree build)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SRC=$ROOT/remote-sensing-vision-language-diffusion-model_amd/csrc
name=$1; shift
mkdir -p $ROOT/tools/ablate
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -fno-gpu-rdc -fno-slp-vectorize "$@" -I$SRC -I$ROOT/include -c $SRC/attention.hip -o /tmp/attn_$name.o 2>/dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/tools/ablate/librsvld_$name.so /tmp/attn_$name.o $SRC/build/conv_igemm.o $SRC/build/conv_halo.o $SRC/build/gemm.o $SRC/build/norm.o $SRC/build/elementwise.o $SRC/build/sampler.o $SRC/build/f32.o $SRC/build/gemv.o $SRC/build/split.o
