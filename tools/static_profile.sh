#!/bin/bash
# CPU-only evidence for a round without GPU access: the static issue budget (tools/isa_cost.py) and the hipcc resource remarks
# of the hot kernels, for the current tree and for an earlier commit.  usage: tools/static_profile.sh <baseline commit> > profiles/rNN_static_isa.txt
set -e
base=${1:-e035223}
root=$(cd "$(dirname "$0")/.." && pwd)
tmp=$(mktemp -d)
flags="--offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only"
emit() {  # <label> <source dir> <extra flags>
  for t in f32 f64; do
    /opt/rocm/bin/hipcc $flags $3 -Rpass-analysis=kernel-resource-usage -o "$tmp/$1_$t.s" "$2/ndzip_amd/csrc/kernels_$t.hip" 2> "$tmp/$1_$t.remarks" || true
  done
}
git -C "$root" archive "$base" ndzip_amd/csrc include | tar -x -C "$tmp"
emit base "$tmp" ""
emit head "$root" "-mllvm -amdgpu-atomic-optimizer-strategy=None"
echo "# static ISA evidence (CPU only; no timing): baseline commit $base vs $(git -C "$root" rev-parse --short HEAD)"
echo "# tools/isa_cost.py: per-iteration ISSUE budget of the persistent kernels' main loops (depth 1) -- see its docstring"
for k in "f32 compress_kernel_dbIfLi3ELb1ELb1E" "f32 compress_kernel_dbIfLi1ELb1ELb0E" "f64 compress_kernel_wideImLi2ELb1E" "f64 compress_kernel_wideImLi3ELb1E"; do
  set -- $k
  for v in base head; do
    echo "## $v  $2"
    python "$root/tools/isa_cost.py" "$tmp/${v}_$1.s" "$2" | sed -n '2,3p;6p'
  done
done
echo "# decompress kernels (one pass per workgroup, no main loop): the whole kernel's issue budget (depth 0)"
for k in "f32 decompress_kernelIfLi3ELb1E" "f32 decompress_kernelIfLi1ELb1E" "f64 decompress_kernelIdLi2ELb1E" "f64 decompress_kernelIdLi3ELb1E"; do
  set -- $k
  for v in base head; do
    echo "## $v  $2"
    python "$root/tools/isa_cost.py" "$tmp/${v}_$1.s" "$2" | sed -n '2p'
  done
done
echo "# f64 decoder with 256 work-items per hypercube (head only; wave-instructions per hypercube = 4 x these, the 128-work-item kernel: 2 x its)"
for k in decompress_kernel_wideILi1ELb1E decompress_kernel_wideILi2ELb1E decompress_kernel_wideILi3ELb1E; do
  echo "## head  $k"
  python "$root/tools/isa_cost.py" "$tmp/head_f64.s" "$k" | sed -n '2p'
done
echo "# hipcc -Rpass-analysis=kernel-resource-usage (head): VGPRs / scratch / occupancy of the codec kernels"
for t in f32 f64; do
  python - "$tmp/head_$t.remarks" <<'PY'
import re, sys
name = None
row = {}
for line in open(sys.argv[1]):
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        name, row = m.group(1), {}
    for key, pat in (("vgprs", r"\bVGPRs: (\d+)"), ("sgprs", r"TotalSGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"),
                     ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"), ("lds", r"LDS Size \[bytes/block\]: (\d+)")):
        m = re.search(pat, line)
        if m and name:
            row[key] = m.group(1)
            if key == "lds" and ("compress_kernel" in name or "decompress_kernel" in name):
                short = re.sub(r"^_ZN9ndzip_hip12_GLOBAL__N_1\d+", "", name)[:40]
                print(f"  {short:40s} VGPRs {row.get('vgprs', '?'):>4s}  SGPRs {row.get('sgprs', '?'):>4s}  scratch {row.get('scratch', '?'):>3s}  waves/SIMD {row.get('occ', '?')}")
PY
done
rm -rf "$tmp"
