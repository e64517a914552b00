#!/usr/bin/env python3
"""Can this compiler's v_bitop3_b32 formation have produced a wrong truth table in a unit?

Found by tools/fuzz_interpreter_vs_compiler.py (round 6, the third compiler finding): AMD LLVM 22.0.0git of ROCm 7.2 fuses and / or /
xor expressions of up to three inputs into gfx950's V_BITOP3_B32 (an 8-bit truth table) -- and computes the table wrongly when an
INNER bitwise value is reached twice from one root, e.g.

    t = A & b;   r = (t & X) | (t ^ b)        ; correct table 0x8c -- the compiler emits 0xac = (A & X) | (~A & b)

(both instruction selectors share the matcher; gfx942, which has no such instruction, is compiled correctly).  The matcher takes `t`
as a source, later replaces that source slot by t's own operand while the bits already computed for the sibling still mean `t`.
Tree-shaped expressions -- every inner value used once inside the fused expression -- are not affected: x ^ (y & z), (x | z) ^ y, ...

This repository's kernels are bit manipulation from end to end, so the question is asked of them twice:
  * statically, here: the optimised LLVM IR of each device unit is searched for the trigger -- a bitwise instruction whose operand
    DAG (through and / or / xor, depth 5) reaches some inner bitwise instruction twice.  The product's units have none
    (tests/test_compiler_sink_audit.py holds that, with REPRODUCER as the positive control); the tables the compiler did emit into
    the product library are listed with a tree-shaped expression each.
  * dynamically, everywhere else: the BUILT code objects are executed against the oracle (tests/test_gfx950_exec.py, the rehearsed
    -m gpu suite, tools/fuzz_code_object.py) -- a wrong minterm in a data path flips bits of the stream, one in an address path moves them.

    python3 tools/audit_bitop3.py                       # the product's units, with the product's flags
    python3 tools/audit_bitop3.py path/to/unit.hip -DNDZIP_EXP_F64_NOCARRY ...
"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"
LLC = "/opt/rocm/lib/llvm/bin/llc"
PRODUCT_FLAGS = ["-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-mllvm", "-amdgpu-atomic-optimizer-strategy=None"]
PRODUCT_UNITS = [os.path.join(ROOT, "ndzip_amd", "csrc", u + ".hip") for u in ("kernels_f32", "kernels_f64", "capi")]

REPRODUCER_IR = """
define i32 @f(i32 %A, i32 %b, i32 %X) {
  %t = and i32 %A, %b
  %u = and i32 %t, %X
  %v = xor i32 %t, %b
  %r = or i32 %u, %v
  ret i32 %r
}
"""
REPRODUCER_TABLE = 0x8C  # with S0 = A = 0xf0, S1 = b = 0xcc, S2 = X = 0xaa: ((0xf0 & 0xcc) & 0xaa) | ((0xf0 & 0xcc) ^ 0xcc)


def scan_ir(text: str, depth: int = 5):
    """[(function, root, [inner values reached twice])] over the LLVM IR text of a module"""
    hits, nroots = [], 0
    for f in re.split(r"\n(?=define )", text):
        if not f.startswith("define"):
            continue
        name = re.match(r"define[^@]*@(\S+?)\(", f).group(1)
        defs = {}
        for m in re.finditer(r"^\s*(%[\w.]+) = (and|or|xor)( disjoint)? (i\d+|<\d+ x i\d+>) ([^,]+), (\S+)", f, re.M):
            defs[m.group(1)] = (m.group(5).strip(), m.group(6).strip())

        def expand(r, d, seen):
            if r not in defs or d == 0:
                return
            seen[r] += 1
            for x in defs[r]:
                expand(x, d - 1, seen)

        for r in defs:
            nroots += 1
            seen = collections.Counter()
            expand(r, depth, seen)
            dup = [x for x, c in seen.items() if c >= 2]
            if dup:
                hits.append((name, r, dup))
    return hits, nroots


def audit(source: str, flags, workdir: str):
    """(hits, bitwise roots looked at) for one HIP unit: its optimised device IR, scanned"""
    ll = os.path.join(workdir, "audit.ll")
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", *flags, "--cuda-device-only", "-emit-llvm", "-S", source, "-o", ll], capture_output=True, text=True)
    if r.returncode:
        raise RuntimeError(f"hipcc failed on {source}:\n{r.stderr[-2000:]}")
    return scan_ir(open(ll).read())


def emitted_table(ir: str, workdir: str, extra=()):
    """the bitop3 table llc emits for a one-function module (None: no v_bitop3 in its code)"""
    p = os.path.join(workdir, "t.ll")
    open(p, "w").write(ir)
    r = subprocess.run([LLC, "-mtriple=amdgcn-amd-amdhsa", "-mcpu=gfx950", "-O3", *extra, p, "-o", "-"], capture_output=True, text=True)
    if r.returncode:
        raise RuntimeError(r.stderr[-1500:])
    m = re.search(r"v_bitop3_b32 .* bitop3:(0x[0-9a-f]+)", r.stdout)
    return int(m.group(1), 16) if m else None


def main(argv):
    sources = [a for a in argv if a.endswith((".hip", ".cu", ".cpp"))] or PRODUCT_UNITS
    flags = PRODUCT_FLAGS + [a for a in argv if not a.endswith((".hip", ".cu", ".cpp"))]
    total = 0
    with tempfile.TemporaryDirectory() as d:
        got = emitted_table(REPRODUCER_IR, d)
        print(f"reproducer: llc emits bitop3:{got:#x}; the expression's table is {REPRODUCER_TABLE:#x} -- {'this compiler has the defect' if got != REPRODUCER_TABLE else 'this compiler is right'}")
        for src in sources:
            hits, nroots = audit(src, flags, d)
            print(f"{os.path.basename(src)}: {nroots} bitwise instructions as roots, {len(hits)} whose operand DAG reaches an inner bitwise value twice")
            for name, root, dup in hits[:20]:
                print(f"    {name[:80]}: {root} reaches {dup} twice")
            total += len(hits)
    print(f"candidates for a wrong v_bitop3 table: {total}")
    return 1 if total else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
