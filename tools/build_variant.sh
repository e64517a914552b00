#!/bin/bash
# Build a variant of the library for A/B work (CPU, no GPU needed): tools/build_variant.sh <name> [--lab] [extra hipcc flags...]
#   -> ndzip_amd/_variants/<name>.so   (git-ignored; travels to the GPU box with gpurun)
# --lab: compile a COPY of ndzip_amd/csrc with tools/experiments/lab_scaffolding.patch applied -- the experiment switches that
# are deliberately not in the product sources (run-time ablation flags and environment knobs, per-phase cycle counters, the
# alternative orderings of the f32 compress iteration, plain-policy input loads).  What they are selected with:
#   -DNDZIP_EXP_KNOBS -DNDZIP_EXP_ABLATION     NDZIP_HIP_EXP / NDZIP_HIP_BPC / NDZIP_HIP_NO_PAIRED (tools/ablate.sh)
#   -DNDZIP_EXP_PHASE_TIMING                   per-phase cycle counters of the f32 compress iteration (NDZIP_HIP_EXP=16)
#   -DNDZIP_EXP_WINDOW_BEHIND_PUBLISH, -DNDZIP_EXP_TRANSPOSE_BEFORE_LOOKBACK, -DNDZIP_EXP_EARLY_VECTORS=n, -DNDZIP_EXP_DB_WAVES=n
#   -DNDZIP_EXP_COPYOUT_BATCH=n                copy-out: the LDS reads of n vectors per work-item issued back to back, one wait, then the stores
#   -DNDZIP_PLAIN_INPUT_LOADS                  default cache policy instead of nt for the read-once input
#   -DNDZIP_EXP_LINEAR_RUN64                   64-bit encoded runs linear in LDS (round-1 layout) instead of XOR-swizzled
#   -DNDZIP_EXP_F64_NOCARRY                    64-bit stencil without borrow chains: x - y of two sums as x + ~y + 1 over v_lshl_add_u64 (variant f64sched)
# e.g. tools/build_variant.sh knobs --lab -DNDZIP_EXP_KNOBS -DNDZIP_EXP_ABLATION
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/ndzip_amd/_variants; mkdir -p "$out/obj_$name"
src=$root/ndzip_amd/csrc
if [ "$1" = "--lab" ]; then
  shift
  src=$out/obj_$name/src/ndzip_amd/csrc; rm -rf "$out/obj_$name/src"; mkdir -p "$src" "$out/obj_$name/src/include"
  cp "$root"/ndzip_amd/csrc/*.hip "$root"/ndzip_amd/csrc/*.hpp "$root"/ndzip_amd/csrc/*.inl "$src"/
  cp "$root"/include/ndzip_hip.h "$out/obj_$name/src/include/"
  patch -s -p1 -d "$src" < "$root/tools/experiments/lab_scaffolding.patch"
fi
for u in kernels_f32 kernels_f64 capi; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wall -Wno-unused-function -mllvm -amdgpu-atomic-optimizer-strategy=None "$@" \
      -c "$src/$u.hip" -o "$out/obj_$name/$u.o" &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$out/$name.so" "$out/obj_$name"/*.o
# this image's compiler can sink an LDS load past __syncthreads() (tools/audit_machine_sink.py): no variant with such a load is built
audit=$(python3 "$root/tools/audit_machine_sink.py" "$src/kernels_f32.hip" "$src/kernels_f64.hip" "$src/capi.hip" "$@") \
  || { echo "$audit" | grep -B1 -A1 "left behind" >&2; echo "build_variant: $name: machine-sink moved a load across a barrier -- variant removed" >&2; rm -f "$out/$name.so"; exit 1; }
echo "$audit" | tail -1
# ... and fuse a bitwise expression with a shared inner value into a wrong v_bitop3 table (tools/audit_bitop3.py)
audit=$(python3 "$root/tools/audit_bitop3.py" "$src/kernels_f32.hip" "$src/kernels_f64.hip" "$src/capi.hip" "$@") \
  || { echo "$audit" | tail -8 >&2; echo "build_variant: $name: a bitwise expression has the shape this compiler fuses wrongly -- variant removed" >&2; rm -f "$out/$name.so"; exit 1; }
echo "$audit" | tail -1
# ... and read the low bytes of a perm result where its sign is asked for (tools/audit_perm_sra.py)
audit=$(python3 "$root/tools/audit_perm_sra.py" "$src/kernels_f32.hip" "$src/kernels_f64.hip" "$src/capi.hip" "$@") \
  || { echo "$audit" | tail -8 >&2; echo "build_variant: $name: an arithmetic byte shift of a perm result -- variant removed" >&2; rm -f "$out/$name.so"; exit 1; }
echo "$audit" | tail -1
echo "$out/$name.so"
