#!/bin/bash
# Build libllada_hip.so of the working tree with extra compiler flags (diagnostic variants for A/B runs through
# LLADA_HIP_LIB): tools/build_variant.sh <out.so> <flags...>
set -e
out=$(realpath -m "$1"); shift
root=$(git -C "$(dirname "$0")" rev-parse --show-toplevel)
tmp=$(mktemp -d)
mkdir -p "$tmp/llada_amd" "$tmp/include"
cp -r "$root/llada_amd/csrc" "$tmp/llada_amd/" && cp "$root"/include/*.h "$tmp/include/" && rm -rf "$tmp/llada_amd/csrc/build"
make -C "$tmp/llada_amd/csrc" -j8 CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=off $*" >/dev/null
cp "$tmp/llada_amd/libllada_hip.so" "$out"
rm -rf "$tmp"
echo "built $out with $*"
