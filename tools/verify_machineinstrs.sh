#!/bin/bash
# Compile the three device translation units with LLVM's machine verifier switched on
# (-verify-machineinstrs): every inline-asm block's operand constraints, clobbers and
# register classes are checked by the compiler that has to schedule around them.
# Device-only, objects go to /tmp; the shipped library is not touched.
set -u
here=$(cd "$(dirname "$0")/.." && pwd)
rc=0
for u in kernels_f32 kernels_f64 capi; do
    out=$(/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden \
        -mllvm -amdgpu-atomic-optimizer-strategy=None -mllvm -verify-machineinstrs \
        --cuda-device-only -c "$here/ndzip_amd/csrc/$u.hip" -o "/tmp/verify_mi_$u.o" 2>&1)
    st=$?
    bad=$(printf '%s\n' "$out" | grep -c -E 'Bad machine code|error:|LLVM ERROR')
    echo "$u: hipcc rc=$st, verifier complaints=$bad"
    [ $st -ne 0 ] || [ "$bad" -ne 0 ] && { printf '%s\n' "$out" | head -20; rc=1; }
done
exit $rc
