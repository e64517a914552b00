#!/bin/bash
# Build a variant of the library for A/B work (CPU, no GPU needed): tools/build_variant.sh <name> [extra hipcc flags...]
#   -> ndzip_amd/_variants/<name>.so   (git-ignored; travels to the GPU box with gpurun)
# e.g. tools/build_variant.sh knobs -DNDZIP_EXP_KNOBS -DNDZIP_EXP_ABLATION   (what tools/ablate.sh needs)
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/ndzip_amd/_variants; mkdir -p "$out/obj_$name"
for u in kernels_f32 kernels_f64 capi; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wall -Wno-unused-function -mllvm -amdgpu-atomic-optimizer-strategy=None "$@" \
      -c "$root/ndzip_amd/csrc/$u.hip" -o "$out/obj_$name/$u.o" &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$out/$name.so" "$out/obj_$name"/*.o
echo "$out/$name.so"
