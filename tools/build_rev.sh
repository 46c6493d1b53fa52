#!/bin/bash
# Build libllada_hip.so of another git revision next to the tree (for in-process A/B timing through
# tools/ab_attn.py / LLADA_HIP_LIB): tools/build_rev.sh <rev> <out.so>
set -e
rev=$1; out=$(realpath -m "$2")
root=$(git -C "$(dirname "$0")" rev-parse --show-toplevel)
tmp=$(mktemp -d)
git -C "$root" archive "$rev" llada_amd/csrc include | tar -x -C "$tmp"
make -C "$tmp/llada_amd/csrc" -j8 >/dev/null
cp "$tmp/llada_amd/libllada_hip.so" "$out"
rm -rf "$tmp"
echo "built $rev -> $out"
