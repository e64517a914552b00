#!/usr/bin/env python3
"""Does a unit shift a v_perm_b32 result arithmetically by whole bytes?  (docs/compiler_findings.md, finding 4)

Found by tools/fuzz_interpreter_vs_compiler.py (round 6): when AMD LLVM 22.0.0git (ROCm 7.2, SelectionDAG; gfx950 and gfx90a alike) turns
a gather of bytes into a v_perm_b32, it looks through the sources of each byte -- and through `ashr x, 8k` with x = __builtin_amdgcn_perm(...)
it asks for byte (index + k) of the perm's selector without noticing that the index has left the word: the bytes that should be copies
of x's sign come out as x's LOW bytes.  REPRODUCER below: 7 lines, every word wrong; GlobalISel compiles it correctly.

The audit: in the optimised IR, an `ashr i32` by 8, 16 or 24 whose operand is the result of a call to llvm.amdgcn.perm.  The product's
kernels call the builtin 2 368 times and shift its results only by `lshr 4` (not a byte multiple: the byte-provider analysis stops there).

    python3 tools/audit_perm_sra.py [unit.hip ...] [flags]
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"
PRODUCT_FLAGS = ["-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-mllvm", "-amdgpu-atomic-optimizer-strategy=None"]
PRODUCT_UNITS = [os.path.join(ROOT, "ndzip_amd", "csrc", u + ".hip") for u in ("kernels_f32", "kernels_f64", "capi")]

REPRODUCER = r"""
#include <hip/hip_runtime.h>
#include <cstdint>
extern "C" __global__ void k_perm(const uint32_t *in, uint32_t *out) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    const uint32_t x = in[i], d = in[i + 512];
    const uint32_t a = __builtin_amdgcn_perm(x, d, 0x020c0006u);     // bytes: x.2, d.0, 0, d.2
    const uint32_t s = (uint32_t) ((int32_t) a >> 24);               // bytes 1..3: copies of the sign of d.2
    out[i] = ((s >> 8) & 0xffu) | (((d >> 24) & 0xffu) << 8) | (((s >> 16) & 0xffu) << 16) | (((d >> 8) & 0xffu) << 24);
}
"""


def scan_ir(text: str):
    """([(function, the ashr)], perm calls, shifts of perm results of any kind) over the LLVM IR text of a module"""
    hits, nperm, nshift = [], 0, 0
    for f in re.split(r"\n(?=define )", text):
        if not f.startswith("define"):
            continue
        name = re.match(r"define[^@]*@(\S+?)\(", f).group(1)
        perms = set(re.findall(r"(%[\w.]+) = (?:tail )?call (?:noundef )?i32 @llvm\.amdgcn\.perm", f))
        nperm += len(perms)
        for m in re.finditer(r"(%[\w.]+) = (ashr|lshr)( exact)? i32 (%[\w.]+), (\d+)", f):
            if m.group(4) in perms:
                nshift += 1
                if m.group(2) == "ashr" and int(m.group(5)) % 8 == 0:
                    hits.append((name, m.group(0).strip()))
    return hits, nperm, nshift


def audit(source: str, flags, workdir: str):
    ll = os.path.join(workdir, "audit.ll")
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", *flags, "--cuda-device-only", "-emit-llvm", "-S", source, "-o", ll], capture_output=True, text=True)
    if r.returncode:
        raise RuntimeError(f"hipcc failed on {source}:\n{r.stderr[-2000:]}")
    return scan_ir(open(ll).read())


def main(argv):
    sources = [a for a in argv if a.endswith((".hip", ".cu", ".cpp"))] or PRODUCT_UNITS
    flags = PRODUCT_FLAGS + [a for a in argv if not a.endswith((".hip", ".cu", ".cpp"))]
    total = 0
    with tempfile.TemporaryDirectory() as d:
        for src in sources:
            hits, nperm, nshift = audit(src, flags, d)
            print(f"{os.path.basename(src)}: {nperm} calls of llvm.amdgcn.perm, {nshift} shifts of their results, {len(hits)} of them arithmetic by whole bytes")
            for name, text in hits[:10]:
                print(f"    {name[:80]}: {text}")
            total += len(hits)
    print(f"arithmetic byte shifts of a perm result: {total}")
    return 1 if total else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
