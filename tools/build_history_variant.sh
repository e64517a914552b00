#!/bin/bash
# Build the library as it was at an earlier commit into ndzip_amd/_variants/<name>.so, for an A/B of a whole step of the history
# (bench.py --lib / tools/ab.sh): tools/build_history_variant.sh <name> <commit> [extra hipcc flags...]
set -e
name=$1; commit=$2; shift 2
root=$(cd "$(dirname "$0")/.." && pwd)
tmp=$(mktemp -d)
git -C "$root" archive "$commit" ndzip_amd/csrc include | tar -x -C "$tmp"
out=$root/ndzip_amd/_variants; mkdir -p "$out"
for u in kernels_f32 kernels_f64 capi; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wall -Wno-unused-function "$@" \
      -c "$tmp/ndzip_amd/csrc/$u.hip" -o "$tmp/$u.o" &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$out/$name.so" "$tmp"/*.o
rm -rf "$tmp"
echo "$out/$name.so"
