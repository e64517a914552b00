#!/usr/bin/env python3
"""Did the compiler move a load across a workgroup barrier?

Found by tools/fuzz_interpreter_vs_compiler.py (round 6): this image's LLVM (ROCm 7.2, hipcc -O1 .. -O3) lets its "Machine code
sinking" pass move an LDS load out of its basic block, past `fence release; s_barrier; fence acquire` (= __syncthreads()), into a
later block when a wave-uniform loop follows and the loaded value is used only after it.  The other wavefronts of the workgroup
are then free to overwrite the LDS location before the load executes: a write-after-read race that is not in the source.  The
smallest program that shows it is REPRODUCER below; the interpreter (tests/gfx950_exec.py) runs it to the wrong answer, as a
device with unlucky timing would.

This tool asks the compiler itself: it compiles each given HIP unit with `-mllvm -print-before=machine-sink -mllvm
-print-after=machine-sink`, and reports every instruction with a memory load that changed basic block while a barrier or fence
stood behind it in its old block.  The product's three device units are clean (tests/test_compiler_sink_audit.py holds that on
every run, next to the reproducer as the positive control); run it over any lab variant's sources before believing a number
measured with it:

    python3 tools/audit_machine_sink.py                       # the product's units, with the product's flags
    python3 tools/audit_machine_sink.py path/to/unit.hip -DNDZIP_EXP_F64_NOCARRY ...
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
extern "C" __global__ void __launch_bounds__(256) k_sink(const uint32_t *in, uint32_t *out) {
    __shared__ uint64_t slot[256];
    const uint32_t t = threadIdx.x, i = blockIdx.x * 256u + t;
    const uint32_t a = in[i], b = in[i + 512];
    slot[(t * 11u + 201u) & 255u] = ((uint64_t) a << 32) | b; __syncthreads();
    const uint64_t first = slot[(t * 60u + 241u) & 255u]; __syncthreads();    // read, THEN the barrier ...
    uint32_t x = a;
    for (uint32_t k = 0, n = (uint32_t) __ballot(b & 1u) & 3u; k < n; ++k) x ^= b >> k;   // a wave-uniform loop: a new basic block
    slot[(t * 15u + 249u) & 255u] = x; __syncthreads();                        // ... after which the slots are written again
    const uint64_t second = slot[(t * 27u + 191u) & 255u]; __syncthreads();
    out[2 * i] = (uint32_t) (first ^ (first >> 32)) + x;
    out[2 * i + 1] = (uint32_t) (second ^ (second >> 32));
}
"""


def _parse(text: str):
    """{function: {'before' | 'after': {block: [instruction text]}}} from the two MIR dumps per function"""
    funcs, cur, phase, block = {}, None, None, None
    for line in text.splitlines():
        if line.startswith("# *** IR Dump"):
            phase, cur = ("before" if "Before" in line else "after"), None
            continue
        m = re.match(r"# Machine code for function (\S+):", line)
        if m:
            cur = funcs.setdefault(m.group(1), {}).setdefault(phase, {})
            block = None
            continue
        if cur is None:
            continue
        m = re.match(r"(bb\.\d+)[^:]*:", line)
        if m and not line.startswith(" "):
            block = m.group(1)
            cur[block] = []
            continue
        if block and line.startswith("  ") and not line.lstrip().startswith(("successors", "liveins", ";")):
            cur[block].append(line.strip())
    return funcs


def _index(blocks):
    out = {}
    for b, ins in blocks.items():
        for k, t in enumerate(ins):
            m = re.match(r"(%\d+):\S+ = ", t)
            if m:
                out[m.group(1)] = (b, k, t)
    return out


def audit(source: str, flags, workdir: str):
    """[(function, instructions sunk, loads among them, [(old block, new block, instruction, [barriers / fences it left behind])])]"""
    obj = os.path.join(workdir, "audit.o")
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", *flags, "--cuda-device-only", "-c", source, "-o", obj,
                        "-mllvm", "-print-before=machine-sink", "-mllvm", "-print-after=machine-sink"], capture_output=True, text=True)
    if r.returncode:
        raise RuntimeError(f"hipcc failed on {source}:\n{r.stderr[-2000:]}")
    rows = []
    for name, ph in _parse(r.stderr).items():
        if "before" not in ph or "after" not in ph:
            continue
        ib, ia = _index(ph["before"]), _index(ph["after"])
        moved = [(reg, ib[reg], ia[reg]) for reg in ib if reg in ia and ib[reg][0] != ia[reg][0]]
        bad, nloads = [], 0
        for reg, (b0, k0, t), (b1, _, _) in moved:
            if not re.search(r"\((volatile |dereferenceable |non-temporal )*load", t) and not re.search(r"\bDS_(READ|BPERMUTE|PERMUTE)", t):
                continue  # (not a load, or an INVARIANT one: the kernel-argument loads, which no barrier orders)
            nloads += 1
            behind = [re.search(r"S_BARRIER|ATOMIC_FENCE", x).group(0) for x in ph["before"][b0][k0 + 1:] if re.search(r"S_BARRIER|ATOMIC_FENCE", x)]
            if behind:
                bad.append((b0, b1, t, behind))
        rows.append((name, len(moved), nloads, bad))
    return rows  # (empty: a unit without device functions -- or a compiler whose pass has another name; callers check)


def main(argv):
    sources = [a for a in argv if a.endswith((".hip", ".cu", ".cpp"))] or PRODUCT_UNITS
    flags = PRODUCT_FLAGS + [a for a in argv if not a.endswith((".hip", ".cu", ".cpp"))]
    total = loads = 0
    with tempfile.TemporaryDirectory() as d:
        for src in sources:
            rows = audit(src, flags, d)
            if not rows:
                print(f"{os.path.basename(src)}: no device function went through machine-sink")
            for name, nmoved, nloads, bad in rows:
                print(f"{os.path.basename(src)}: {name[:90]}: {nmoved} instructions sunk to another block, {nloads} loads among them, {len(bad)} past a barrier")
                loads += nloads
                for b0, b1, t, behind in bad:
                    print(f"    {b0} -> {b1}: {t[:130]}\n        left behind in {b0}: {behind}")
                total += len(bad)
    print(f"loads moved to another block by machine-sink: {loads}; across a workgroup barrier: {total}")
    return 1 if total else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
