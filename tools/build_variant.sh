#!/bin/bash
# tools/build_variant.sh <name> <file.hip> <hipcc flags...>  ->  tools/ablate/librsvld_<name>.so : <file.hip> rebuilt with the flags
# (-D switches of the kernels' experiment macros), every other object taken from the in-tree build.  For tools/ab.sh.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SRC=$ROOT/remote-sensing-vision-language-diffusion-model_amd/csrc
name=$1; file=$2; shift 2
stem=$(basename "$file" .hip)
mkdir -p "$ROOT/tools/ablate"
extra=""; { [ "$stem" = attention ] || [ "$stem" = split ]; } && extra=-fno-slp-vectorize
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -fno-gpu-rdc $extra "$@" -I"$SRC" -I"$ROOT/include" -c "$SRC/$stem.hip" -o "/tmp/${stem}_$name.o"
objs=""
for o in "$SRC"/build/*.o; do [ "$(basename "$o" .o)" = "$stem" ] || objs="$objs $o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/tools/ablate/librsvld_$name.so" "/tmp/${stem}_$name.o" $objs
echo "built tools/ablate/librsvld_$name.so"
