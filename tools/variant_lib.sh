#!/bin/bash
# Build a variant of libst_hip.so with extra -D flags for ONE source file (kernel experiments):
#   tools/variant_lib.sh <tag> <source.hip> [-DX=Y ...]   ->  exp/libst_hip_<tag>.so
# The other objects come from style_transfer_amd/build/ (run `python -m style_transfer_amd.build` first).
# On the GPU box: cp exp/libst_hip_<tag>.so style_transfer_amd/libst_hip.so (the box works on a scratch copy).
set -e
tag=$1; src=$2; shift 2
root=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$root/exp"
obj="$root/exp/$(basename "$src").$tag.o"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I "$root/include" -I "$root/style_transfer_amd/csrc" \
  -Wall -Wno-unused-function "$@" -x hip -c "$root/style_transfer_amd/csrc/$src" -o "$obj"
others=$(ls "$root"/style_transfer_amd/build/*.o | grep -v "/$src.o$")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $others "$obj" -o "$root/exp/libst_hip_$tag.so"
echo "$root/exp/libst_hip_$tag.so"
