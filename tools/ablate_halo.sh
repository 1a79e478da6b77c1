#!/bin/bash
# Diagnostic builds of the library with parts of conv_halo32_kernel removed (results WRONG by construction; timing only)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SRC=$ROOT/remote-sensing-vision-language-diffusion-model_amd/csrc
mkdir -p $ROOT/tools/ablate
for N in 1 2 3; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -fno-gpu-rdc -DHALO_ABL=$N -I$SRC -I$ROOT/include -c $SRC/conv_halo.hip -o /tmp/halo_abl$N.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/tools/ablate/librsvld_habl$N.so /tmp/halo_abl$N.o $SRC/build/conv_igemm.o $SRC/build/gemm.o $SRC/build/norm.o $SRC/build/attention.o $SRC/build/elementwise.o $SRC/build/sampler.o $SRC/build/f32.o $SRC/build/gemv.o
done
ls $ROOT/tools/ablate
