"""LLVM as the outside authority for the bit-level primitives the kernels are built from -- v_perm_b32 (the transposes' byte shuffles),
v_alignbit_b32 (rotl1 / rotr1, 64-bit rotates), v_bfe_i32 / v_bfe_u32 (the decoders' plane walk), v_bitop3_b32 (complement, swizzle,
keep-mask) -- read two ways, neither of them this repository's:

  * constant folding: `opt -passes=instcombine` evaluates llvm.amdgcn.perm / llvm.fshr / llvm.amdgcn.sbfe / ubfe on constant operands
    (the compiler writers' executable statement of what the instruction computes);
  * instruction selection: `llc -mcpu=gfx950` turns a three-input boolean expression into ONE v_bitop3_b32 with a truth-table immediate
    and an operand order of its choosing.

Both emulators the CPU suite trusts must agree with that: the functional model's __builtin_amdgcn_perm / _alignbit
(tests/wavesim/hip/hip_runtime.h) and the instruction-level interpreter's opcode semantics (tests/gfx950_exec.py).  Rounds 3-4 showed
that a reading shared by the kernels' author and the emulators' author passes every check they wrote themselves."""
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
pytestmark = pytest.mark.skipif(not os.path.exists(os.path.join(LLVM, "opt")), reason="needs the ROCm LLVM tools")


def _fold(functions):
    """{name: folded i32} of `define i32 @name() { ret <call on constants> }` bodies, by LLVM's InstCombine."""
    ir = ['target triple = "amdgcn-amd-amdhsa"', "declare i32 @llvm.amdgcn.perm(i32, i32, i32)", "declare i32 @llvm.fshr.i32(i32, i32, i32)",
          "declare i32 @llvm.amdgcn.sbfe.i32(i32, i32, i32)", "declare i32 @llvm.amdgcn.ubfe.i32(i32, i32, i32)"]
    for name, call in functions.items():
        ir.append(f"define i32 @{name}() {{\n  %r = call i32 {call}\n  ret i32 %r\n}}")
    r = subprocess.run([os.path.join(LLVM, "opt"), "-passes=instcombine", "-S", "-"], input="\n".join(ir), capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    out = {}
    for m in re.finditer(r"define i32 @(\w+)\(\)[^{]*\{\s*ret i32 (-?\d+)\s*\}", r.stdout):
        out[m.group(1)] = int(m.group(2)) & 0xFFFFFFFF
    assert set(out) == set(functions), f"LLVM did not fold {sorted(set(functions) - set(out))[:5]}"
    return out


def _s(x):
    """an i32 literal"""
    x &= 0xFFFFFFFF
    return str(x - (1 << 32) if x >= 1 << 31 else x)


class _Wave:
    def __init__(self, regs):
        self.regs, self.out = regs, None

    def rv32(self, a):
        return self.regs[a]

    def wv32(self, d, v, mask=None):
        self.out = np.broadcast_to(np.asarray(v, dtype=np.uint32), (64,)).copy()


class _Ins:
    def __init__(self, op, nsrc, mods=None):
        self.op, self.args, self.mods, self.text = op, ["d"] + [f"s{k}" for k in range(nsrc)], mods or {}, op


def _interp(op, srcs, mods=None):
    """the interpreter's semantics of `op` on 64 lanes of operands"""
    from tests import gfx950_exec as gx

    w = _Wave({f"s{k}": np.asarray(v, dtype=np.uint32) for k, v in enumerate(srcs)})
    gx.OPS[op](w, _Ins(op, len(srcs), mods))
    return w.out


def _cases(seed, n):
    rng = np.random.default_rng(seed)
    a = rng.integers(0, 1 << 32, size=n, dtype=np.uint64).astype(np.uint32)
    b = rng.integers(0, 1 << 32, size=n, dtype=np.uint64).astype(np.uint32)
    return a, b, rng


def _model_values(tmp_path, triples, what):
    """the functional model's __builtin_amdgcn_perm / __builtin_amdgcn_alignbit on the same operands (host C++ against its header)"""
    src = tmp_path / f"model_{what}.cc"
    rows = ",\n".join(f"{{{a}u, {b}u, {c}u}}" for a, b, c in triples)
    src.write_text(f'#include <cstdio>\n#include <hip/hip_runtime.h>\nstatic const uint32_t T[][3] = {{\n{rows}\n}};\n'
                   f'int main() {{ for (auto &t : T) printf("%u\\n", (unsigned) __builtin_amdgcn_{what}(t[0], t[1], t[2])); }}\n')
    exe = tmp_path / f"model_{what}"
    r = subprocess.run([os.path.join(LLVM, "clang++"), "-std=c++17", "-O1", "-Wno-unknown-attributes", "-I", os.path.join(ROOT, "tests", "wavesim"), str(src), "-o", str(exe),
                        "-pthread"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    return [int(x) for x in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split()]


def test_v_perm_b32_is_what_llvm_folds(tmp_path):
    a, b, rng = _cases(1, 64)
    sels = []
    for k in range(16):                       # every selector value in every byte position (0-3 src1, 4-7 src0, 8-11 sign fills, 12 zero, 13+ ones)
        sels += [k | (k << 8) | (k << 16) | (k << 24), k | ((15 - k) << 8) | (((k + 5) % 16) << 16) | (((k + 9) % 16) << 24)]
    sels += [0x07060504, 0x03020100, 0x05010400, 0x07030602, 0x06040200, 0x07050301]   # the byte picks of a 32x32 transpose network
    sels += [int(x) for x in rng.integers(0, 16, size=(26, 4), dtype=np.int64) @ np.array([1, 1 << 8, 1 << 16, 1 << 24])]
    sels = np.array(sels[:64], dtype=np.uint32)
    want = _fold({f"p{i}": f"@llvm.amdgcn.perm(i32 {_s(int(a[i]))}, i32 {_s(int(b[i]))}, i32 {_s(int(sels[i]))})" for i in range(64)})
    llvm = np.array([want[f"p{i}"] for i in range(64)], dtype=np.uint32)
    assert np.array_equal(_interp("v_perm_b32", [a, b, sels]), llvm), "the interpreter's v_perm_b32 disagrees with LLVM's constant folding"
    model = np.array(_model_values(tmp_path, list(zip(a.tolist(), b.tolist(), sels.tolist())), "perm"), dtype=np.uint32)
    assert np.array_equal(model, llvm), "the functional model's __builtin_amdgcn_perm disagrees with LLVM's constant folding"


def test_v_alignbit_b32_is_llvm_fshr(tmp_path):
    a, b, rng = _cases(2, 64)
    sh = np.array(list(range(32)) + [31, 1, 0, 32, 33, 63, 64, 95] + [int(x) for x in rng.integers(0, 256, size=24)], dtype=np.uint32)
    want = _fold({f"f{i}": f"@llvm.fshr.i32(i32 {_s(int(a[i]))}, i32 {_s(int(b[i]))}, i32 {_s(int(sh[i]))})" for i in range(64)})
    llvm = np.array([want[f"f{i}"] for i in range(64)], dtype=np.uint32)
    assert np.array_equal(_interp("v_alignbit_b32", [a, b, sh]), llvm)        # v_alignbit_b32 D, hi, lo, shift == fshr(hi, lo, shift)
    model = np.array(_model_values(tmp_path, list(zip(a.tolist(), b.tolist(), sh.tolist())), "alignbit"), dtype=np.uint32)
    assert np.array_equal(model, llvm)


def test_v_bfe_is_what_llvm_folds():
    a, _, rng = _cases(3, 64)
    off = np.array([int(x) for x in rng.integers(0, 32, size=64)], dtype=np.uint32)
    width = np.array([1] * 16 + [int(x) for x in rng.integers(0, 32, size=40)] + [0, 31, 32, 33, 1, 8, 16, 24], dtype=np.uint32)
    off[:16] = np.arange(31, 15, -1)            # the decoders' walk: one head bit at a time, from the top
    keep = (off.astype(np.int64) + (width & 31)) <= 32   # (LLVM folds the in-range cases; out-of-range fields are not used by the kernels)
    fs = {}
    for i in np.flatnonzero(keep):
        fs[f"s{i}"] = f"@llvm.amdgcn.sbfe.i32(i32 {_s(int(a[i]))}, i32 {int(off[i])}, i32 {int(width[i])})"
        fs[f"u{i}"] = f"@llvm.amdgcn.ubfe.i32(i32 {_s(int(a[i]))}, i32 {int(off[i])}, i32 {int(width[i])})"
    want = _fold(fs)
    si, ui = _interp("v_bfe_i32", [a, off, width]), _interp("v_bfe_u32", [a, off, width])
    for i in np.flatnonzero(keep):
        assert int(si[i]) == want[f"s{i}"] and int(ui[i]) == want[f"u{i}"], (hex(int(a[i])), int(off[i]), int(width[i]))


EXPRESSIONS = {  # name: (LLVM IR body over %a %b %c -> %r, the same in numpy)
    "and_xor": ("%t = and i32 %a, %b\n  %r = xor i32 %t, %c", lambda a, b, c: (a & b) ^ c),
    "or_andn": ("%n = xor i32 %c, -1\n  %t = and i32 %b, %n\n  %r = or i32 %a, %t", lambda a, b, c: a | (b & ~c)),
    "xor_and": ("%t = xor i32 %a, %b\n  %r = and i32 %t, %c", lambda a, b, c: (a ^ b) & c),
    "nor_xor": ("%t = or i32 %a, %b\n  %n = xor i32 %t, -1\n  %r = xor i32 %n, %c", lambda a, b, c: ~(a | b) ^ c),
    "swizzle": ("%t = and i32 %a, %c\n  %r = xor i32 %b, %t", lambda a, b, c: b ^ (a & c)),                       # at(a) of run_layout: a ^ ((a >> 3) & 0x70)
    "keep_complement": ("%t = and i32 %a, %b\n  %r = xor i32 %t, %c", lambda a, b, c: (a & b) ^ c),                # decoder: (word & kept) ^ sign plane
    "select": ("%t = and i32 %a, %b\n  %n = xor i32 %a, -1\n  %u = and i32 %n, %c\n  %r = or i32 %t, %u", lambda a, b, c: (a & b) | (~a & c)),
    "maj": ("%t = and i32 %a, %b\n  %u = and i32 %a, %c\n  %v = and i32 %b, %c\n  %w = or i32 %t, %u\n  %r = or i32 %w, %v", lambda a, b, c: (a & b) | (a & c) | (b & c)),
    "xor3": ("%t = xor i32 %a, %b\n  %r = xor i32 %t, %c", lambda a, b, c: a ^ b ^ c),
    "complement_negative": ("%t = and i32 %a, %b\n  %r = xor i32 %c, %t", lambda a, b, c: c ^ (a & b)),            # v ^ (sign mask & 0x7fffffff)
    "andn_or_and": ("%n = xor i32 %a, -1\n  %t = and i32 %n, %b\n  %u = and i32 %a, %c\n  %v = xor i32 %t, %u\n  %r = xor i32 %v, -1", lambda a, b, c: ~((~a & b) ^ (a & c))),
}


def test_v_bitop3_b32_truth_tables_are_read_the_way_llvm_writes_them(tmp_path):
    ir = ['target triple = "amdgcn-amd-amdhsa"']
    for name, (body, _) in EXPRESSIONS.items():
        ir.append(f"define i32 @{name}(i32 %a, i32 %b, i32 %c) {{\n  {body}\n  ret i32 %r\n}}")
    r = subprocess.run([os.path.join(LLVM, "llc"), "-mtriple=amdgcn-amd-amdhsa", "-mcpu=gfx950", "-O2", "-o", "-"], input="\n".join(ir), capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    rng = np.random.default_rng(4)
    args = {f"v{k}": rng.integers(0, 1 << 32, size=64, dtype=np.uint64).astype(np.uint32) for k in range(3)}  # arguments arrive in v0, v1, v2
    bitop3, bfi = 0, 0
    for name, (_, fn) in EXPRESSIONS.items():
        body = re.search(rf"^{name}:.*?s_setpc_b64", r.stdout, re.S | re.M).group(0)
        valu = [l.split(";")[0].strip() for l in body.splitlines() if l.strip().startswith("v_")]
        # whatever LLVM selected -- one v_bitop3_b32 with a table and an operand order of its choosing, or v_xor + v_bfi_b32 for the
        # majority function -- executed instruction by instruction with the INTERPRETER's semantics must give the expression's value
        regs = dict(args)
        for ins in valu:
            m = re.fullmatch(r"(v_\w+) (v\d), (v\d), (v\d)(?:, (v\d))?(?: bitop3:(0x[0-9a-f]+|\d+))?", ins)
            assert m, ins
            op, dst, srcs = m.group(1), m.group(2), [x for x in m.group(3, 4, 5) if x]
            regs[dst] = _interp(op, [regs[x] for x in srcs], {"bitop3": m.group(6)} if m.group(6) else None)
            bitop3 += op == "v_bitop3_b32"
            bfi += op == "v_bfi_b32"
        assert np.array_equal(regs["v0"], fn(args["v0"], args["v1"], args["v2"]).astype(np.uint32)), (name, valu)
    assert bitop3 >= 8 and bfi >= 1, f"LLVM selected v_bitop3_b32 {bitop3} times and v_bfi_b32 {bfi} times: the comparison has lost its subject"


ARITHMETIC = {  # fused integer operations the kernels' address and offset arithmetic compiles to
    "lshl_add": ("%t = shl i32 %a, 3\n  %r = add i32 %t, %b", lambda a, b, c: (a << np.uint32(3)) + b),
    "add_lshl": ("%t = add i32 %a, %b\n  %r = shl i32 %t, 2", lambda a, b, c: (a + b) << np.uint32(2)),
    "lshl_or": ("%t = shl i32 %a, 5\n  %r = or i32 %t, %b", lambda a, b, c: (a << np.uint32(5)) | b),
    "add3": ("%t = add i32 %a, %b\n  %r = add i32 %t, %c", lambda a, b, c: a + b + c),
    "xad": ("%t = xor i32 %a, %b\n  %r = add i32 %t, %c", lambda a, b, c: (a ^ b) + c),
    "sub": ("%r = sub i32 %a, %b", lambda a, b, c: a - b),
    "lshr_var": ("%s = and i32 %b, 31\n  %r = lshr i32 %a, %s", lambda a, b, c: a >> (b & np.uint32(31))),
    "ashr_31": ("%r = ashr i32 %a, 31", lambda a, b, c: (a.view(np.int32) >> 31).view(np.uint32)),
    "mul_lo": ("%r = mul i32 %a, %b", lambda a, b, c: (a.astype(np.uint64) * b.astype(np.uint64)).astype(np.uint32)),
    "umin": ("%k = icmp ult i32 %a, %b\n  %r = select i1 %k, i32 %a, i32 %b", lambda a, b, c: np.minimum(a, b)),
    "bfe_u": ("%t = lshr i32 %a, 7\n  %r = and i32 %t, 31", lambda a, b, c: (a >> np.uint32(7)) & np.uint32(31)),
    "bfe_i": ("%t = shl i32 %a, 13\n  %r = ashr i32 %t, 31", lambda a, b, c: ((a << np.uint32(13)).view(np.int32) >> 31).view(np.uint32)),  # the plane walk's 0 / -1 per head bit
    "rotl1": ("%h = shl i32 %a, 1\n  %l = lshr i32 %a, 31\n  %r = or i32 %h, %l", lambda a, b, c: (a << np.uint32(1)) | (a >> np.uint32(31))),
}


def test_fused_integer_operations_as_llvm_selects_them():
    """The same idea for the arithmetic around the bit work: LLVM's selection for gfx950 of small integer expressions (v_lshl_add_u32,
    v_add_lshl_u32, v_lshl_or_b32, v_add3_u32, v_xad_u32, v_bfe_*, v_alignbit_b32 for a rotate, ...), executed with the interpreter's
    operand order and semantics, must give the expression's value."""
    ir = ['target triple = "amdgcn-amd-amdhsa"']
    for name, (body, _) in ARITHMETIC.items():
        ir.append(f"define i32 @{name}(i32 %a, i32 %b, i32 %c) {{\n  {body}\n  ret i32 %r\n}}")
    r = subprocess.run([os.path.join(LLVM, "llc"), "-mtriple=amdgcn-amd-amdhsa", "-mcpu=gfx950", "-O2", "-o", "-"], input="\n".join(ir), capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    rng = np.random.default_rng(5)
    args = {f"v{k}": rng.integers(0, 1 << 32, size=64, dtype=np.uint64).astype(np.uint32) for k in range(3)}
    seen = set()
    for name, (_, fn) in ARITHMETIC.items():
        body = re.search(rf"^{name}:.*?s_setpc_b64", r.stdout, re.S | re.M).group(0)
        regs = dict(args)
        for ins in [l.split(";")[0].strip() for l in body.splitlines() if l.strip().startswith("v_")]:
            op, rest = ins.split(None, 1)
            toks = [t.strip() for t in rest.split(",")]
            srcs = [regs[t] if t in regs else np.full(64, int(t, 0) & 0xFFFFFFFF, dtype=np.uint32) for t in toks[1:]]
            regs[toks[0]] = _interp(op, srcs)
            seen.add(op.replace("_e32", "").replace("_e64", ""))
        with np.errstate(over="ignore"):
            want = fn(args["v0"], args["v1"], args["v2"]).astype(np.uint32)
        assert np.array_equal(regs["v0"], want), (name, body)
    for op in ("v_lshl_add_u32", "v_add_lshl_u32", "v_lshl_or_b32", "v_add3_u32", "v_xad_u32", "v_bfe_u32", "v_bfe_i32", "v_alignbit_b32"):
        assert op in seen, f"LLVM no longer selects {op} here: {sorted(seen)}"


# ---- whole micro-functions: llc's gfx950 code for integer IR, executed by the interpreter on a real Wave -------------------------------

M64 = (1 << 64) - 1


def _u64(x):
    return np.asarray(x, dtype=np.uint64)


WIDE = {  # name: (signature, IR body -> %r, numpy on uint64 x, y [and uint32 a, b]) ; i64 arguments arrive in v[0:1], v[2:3], the result leaves in v[0:1]
    "add64": ("i64 %x, i64 %y", "i64", "%r = add i64 %x, %y", lambda x, y: x + y),
    "sub64": ("i64 %x, i64 %y", "i64", "%r = sub i64 %x, %y", lambda x, y: x - y),                       # v_sub_co_u32 / v_subb_co_u32: the f64 stencil's borrow chain
    "lshl_add64": ("i64 %x, i64 %y", "i64", "%t = shl i64 %x, 3\n  %r = add i64 %t, %y", lambda x, y: (x << np.uint64(3)) + y),
    "diff_of_sums": ("i64 %x, i64 %y", "i64", "%n = xor i64 %y, -1\n  %t = add i64 %x, %n\n  %r = add i64 %t, 1", lambda x, y: x + (~y) + np.uint64(1)),
    "rotl1_64": ("i64 %x, i64 %y", "i64", "%h = shl i64 %x, 1\n  %l = lshr i64 %x, 63\n  %r = or i64 %h, %l", lambda x, y: (x << np.uint64(1)) | (x >> np.uint64(63))),
    "rotr1_64": ("i64 %x, i64 %y", "i64", "%h = lshr i64 %x, 1\n  %l = shl i64 %x, 63\n  %r = or i64 %h, %l", lambda x, y: (x >> np.uint64(1)) | (x << np.uint64(63))),
    "shl64_var": ("i64 %x, i64 %y", "i64", "%s = and i64 %y, 63\n  %r = shl i64 %x, %s", lambda x, y: x << (y & np.uint64(63))),
    "lshr64_var": ("i64 %x, i64 %y", "i64", "%s = and i64 %y, 63\n  %r = lshr i64 %x, %s", lambda x, y: x >> (y & np.uint64(63))),
    "ashr64_63": ("i64 %x, i64 %y", "i64", "%r = ashr i64 %x, 63", lambda x, y: (x.view(np.int64) >> np.int64(63)).view(np.uint64)),
    "complement_negative64": ("i64 %x, i64 %y", "i64", "%m = ashr i64 %x, 63\n  %k = and i64 %m, 9223372036854775807\n  %r = xor i64 %x, %k",
                              lambda x, y: x ^ ((x.view(np.int64) >> np.int64(63)).view(np.uint64) & np.uint64(0x7FFFFFFFFFFFFFFF))),
    "and_or_xor64": ("i64 %x, i64 %y", "i64", "%a = and i64 %x, %y\n  %o = or i64 %x, %y\n  %r = xor i64 %a, %o", lambda x, y: (x & y) ^ (x | y)),
    "umin64": ("i64 %x, i64 %y", "i64", "%k = icmp ult i64 %x, %y\n  %r = select i1 %k, i64 %x, i64 %y", lambda x, y: np.minimum(x, y)),
    "smax64": ("i64 %x, i64 %y", "i64", "%k = icmp sgt i64 %x, %y\n  %r = select i1 %k, i64 %x, i64 %y", lambda x, y: np.maximum(x.view(np.int64), y.view(np.int64)).view(np.uint64)),
    "mul64": ("i64 %x, i64 %y", "i64", "%r = mul i64 %x, %y", lambda x, y: x * y),
    "mulhi32": ("i64 %x, i64 %y", "i64", "%a = and i64 %x, 4294967295\n  %b = and i64 %y, 4294967295\n  %m = mul i64 %a, %b\n  %r = lshr i64 %m, 32",
                lambda x, y: ((x & np.uint64(0xFFFFFFFF)) * (y & np.uint64(0xFFFFFFFF))) >> np.uint64(32)),   # the tile origin's magic-number division
    "mad_u64_u32": ("i64 %x, i64 %y", "i64", "%a = and i64 %x, 4294967295\n  %b = lshr i64 %x, 32\n  %m = mul i64 %a, %b\n  %r = add i64 %m, %y",
                    lambda x, y: (x & np.uint64(0xFFFFFFFF)) * (x >> np.uint64(32)) + y),
    "popcount": ("i64 %x, i64 %y", "i64", "%c = call i64 @llvm.ctpop.i64(i64 %x)\n  %r = add i64 %c, %y",
                 lambda x, y: np.array([bin(int(v)).count("1") for v in x], dtype=np.uint64) + y),                  # the decoders' popcount offsets
    "eq_select": ("i64 %x, i64 %y", "i64", "%k = icmp eq i64 %x, 0\n  %r = select i1 %k, i64 %y, i64 %x", lambda x, y: np.where(x == 0, y, x)),
}


def _run_llc_function(asm_body, regs64):
    """Execute the straight-line gfx950 code llc made of one function on a Wave of the interpreter; returns the Wave."""
    from tests import gfx950_exec as gx

    w = gx.Wave(None, 0, {}, 0, "llc")
    for k, val in regs64.items():  # {first VGPR of the pair: uint64[64]}
        w.v[k] = (val & np.uint64(0xFFFFFFFF)).astype(np.uint32)
        w.v[k + 1] = (val >> np.uint64(32)).astype(np.uint32)
    executed = []
    for line in asm_body.splitlines():
        line = line.split(";")[0].strip()
        if not line or line.endswith(":") or line.startswith("."):
            continue
        op, _, rest = line.partition(" ")
        if op == "s_setpc_b64":
            break
        args, mods = gx._split_operands(rest.strip())
        ins = gx.Ins(op, args, mods, 0, 4, line)
        if op not in gx.OPS:
            raise KeyError(op)
        gx.OPS[op](w, ins)
        executed.append(op)
    return w, executed


def test_integer_micro_functions_compiled_by_llc_run_right_on_the_interpreter():
    """Wider than single opcodes: 64-bit adds and borrow chains, shifts by a register, rotates, compares and selects, 32 x 32 -> 64
    multiplies, multiply-adds and population counts -- each a small IR function compiled by llc for gfx950 and executed, VCC and all, by
    the interpreter that runs the BUILT kernels in this suite.  The values are checked against plain 64-bit arithmetic."""
    ir = ['target triple = "amdgcn-amd-amdhsa"', "declare i64 @llvm.ctpop.i64(i64)"]
    for name, (sig, ret, body, _) in WIDE.items():
        ir.append(f"define {ret} @{name}({sig}) {{\n  {body}\n  ret {ret} %r\n}}")
    r = subprocess.run([os.path.join(LLVM, "llc"), "-mtriple=amdgcn-amd-amdhsa", "-mcpu=gfx950", "-O2", "-o", "-"], input="\n".join(ir), capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    rng = np.random.default_rng(6)
    x = rng.integers(0, 1 << 64, size=64, dtype=np.uint64)
    y = rng.integers(0, 1 << 64, size=64, dtype=np.uint64)
    x[:6] = [0, 1, M64, 1 << 63, (1 << 63) - 1, 1 << 32]            # carries, borrows and sign edges
    y[:6] = [0, M64, 1, 1, 1 << 63, (1 << 32) - 1]
    seen, skipped = set(), []
    for name, (_, _, _, fn) in WIDE.items():
        body = re.search(rf"^{name}:.*?s_setpc_b64[^\n]*", r.stdout, re.S | re.M).group(0)
        try:
            w, executed = _run_llc_function(body.split("\n", 1)[1], {0: x, 2: y})
        except KeyError as e:  # (an opcode the kernels never use and the interpreter therefore does not know: not this test's subject)
            skipped.append((name, str(e)))
            continue
        seen.update(o.replace("_e32", "").replace("_e64", "") for o in executed)
        got = w.v[0].astype(np.uint64) | (w.v[1].astype(np.uint64) << np.uint64(32))
        with np.errstate(over="ignore"):
            want = _u64(fn(x, y))
        assert np.array_equal(got, want), (name, body)
    assert len(skipped) <= 3, skipped
    for op in ("v_lshl_add_u64", "v_sub_co_u32", "v_subb_co_u32", "v_alignbit_b32", "v_lshlrev_b64", "v_lshrrev_b64", "v_mad_u64_u32", "v_cndmask_b32", "v_bcnt_u32_b32"):
        assert op in seen, f"{op} was not exercised: {sorted(seen)}; skipped {skipped}"


M32 = (1 << 32) - 1


def _sx(v, bits):
    return v - (1 << bits) if v >> (bits - 1) else v


SCALAR = {  # uniform arithmetic (the tile origin, running pointers, lengths): (IR body over i32 %a %b, i64 %z %w -> i64 %r, the same on Python ints)
    "s_mad": ("%x = zext i32 %a to i64\n  %y = zext i32 %b to i64\n  %m = mul i64 %x, %y\n  %r = add i64 %m, %z", lambda a, b, z, w: a * b + z),
    "s_udiv12": ("%q = udiv i32 %a, 12\n  %r = zext i32 %q to i64", lambda a, b, z, w: a // 12),               # magic-number division
    "s_udiv_row": ("%q = udiv i32 %a, 510\n  %m = urem i32 %a, 510\n  %t = mul i32 %q, %b\n  %u = add i32 %t, %m\n  %r = zext i32 %u to i64",
                   lambda a, b, z, w: ((a // 510) * b + a % 510) & M32),
    "s_add64": ("%r = add i64 %z, %w", lambda a, b, z, w: z + w),
    "s_sub64": ("%r = sub i64 %z, %w", lambda a, b, z, w: z - w),
    "s_shl64": ("%s = and i32 %a, 63\n  %e = zext i32 %s to i64\n  %r = shl i64 %z, %e", lambda a, b, z, w: z << (a & 63)),
    "s_lshr64": ("%s = and i32 %a, 63\n  %e = zext i32 %s to i64\n  %r = lshr i64 %z, %e", lambda a, b, z, w: z >> (a & 63)),
    "s_andn2": ("%n = xor i64 %w, -1\n  %r = and i64 %z, %n", lambda a, b, z, w: z & ~w),
    "s_orn2": ("%n = xor i64 %w, -1\n  %r = or i64 %z, %n", lambda a, b, z, w: z | (~w & M64)),
    "s_umin": ("%k = icmp ult i32 %a, %b\n  %m = select i1 %k, i32 %a, i32 %b\n  %r = zext i32 %m to i64", lambda a, b, z, w: min(a, b)),
    "s_smax": ("%k = icmp sgt i32 %a, %b\n  %m = select i1 %k, i32 %a, i32 %b\n  %r = zext i32 %m to i64", lambda a, b, z, w: max(_sx(a, 32), _sx(b, 32)) & M32),
    "s_cselect": ("%k = icmp uge i32 %a, %b\n  %r = select i1 %k, i64 %z, i64 %w", lambda a, b, z, w: z if a >= b else w),
    "s_lshl_add": ("%t = shl i32 %a, 2\n  %u = add i32 %t, %b\n  %r = zext i32 %u to i64", lambda a, b, z, w: ((a << 2) + b) & M32),
    "s_bfe": ("%t = lshr i32 %a, 5\n  %u = and i32 %t, 1023\n  %r = zext i32 %u to i64", lambda a, b, z, w: (a >> 5) & 1023),
    "s_mulhi": ("%x = zext i32 %a to i64\n  %y = zext i32 %b to i64\n  %m = mul i64 %x, %y\n  %r = lshr i64 %m, 32", lambda a, b, z, w: (a * b) >> 32),
    "s_ashr": ("%x = ashr i64 %z, 63\n  %r = xor i64 %x, %w", lambda a, b, z, w: ((M64 if z >> 63 else 0) ^ w)),
}


def test_uniform_integer_functions_compiled_by_llc_run_right_on_the_interpreter():
    """The scalar side the same way: llc's SALU code for gfx950 (s_mul_i32 / s_mul_hi_u32 magic divisions, s_add_u32 / s_addc_u32 and
    s_sub_u32 / s_subb_u32 through SCC, 64-bit shifts, s_andn2 / s_orn2, s_min / s_max, s_cmp + s_cselect, s_lshlN_add_u32, s_bfe)
    on the interpreter's scalar state.  (Shader calling convention: inreg arguments arrive in s2.., the result leaves in s[0:1].)"""
    from tests import gfx950_exec as gx

    ir = ['target triple = "amdgcn--mesa3d"']
    for name, (body, _) in SCALAR.items():
        ir.append(f"define amdgpu_cs inreg i64 @{name}(i32 inreg %a, i32 inreg %b, i64 inreg %z, i64 inreg %w) {{\n  {body}\n  ret i64 %r\n}}")
    r = subprocess.run([os.path.join(LLVM, "llc"), "-mtriple=amdgcn--mesa3d", "-mcpu=gfx950", "-O2", "-o", "-"], input="\n".join(ir), capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    rng = np.random.default_rng(7)
    tuples = [(0, 0, 0, 0), (M32, M32, M64, M64), (1, M32, M64, 1), (1 << 31, 1, 1 << 63, (1 << 63) - 1), (509, 510, 1 << 32, M32), (510, 4096, M32, 1 << 32)]
    tuples += [(int(rng.integers(0, 1 << 32)), int(rng.integers(0, 1 << 32)), int(rng.integers(0, 1 << 64, dtype=np.uint64)), int(rng.integers(0, 1 << 64, dtype=np.uint64)))
               for _ in range(34)]
    seen, skipped = set(), []
    names = list(SCALAR)
    for k, name in enumerate(names):
        nxt = names[k + 1] if k + 1 < len(names) else None
        m = re.search(rf"^{name}:(.*?)(?=^{nxt}:)" if nxt else rf"^{name}:(.*)", r.stdout, re.S | re.M)
        body = [l.split(";")[0].strip() for l in m.group(1).splitlines()]
        body = [l for l in body if l and not l.endswith(":") and not l.startswith(".") and l.split()[0].startswith(("s_", "v_"))]
        if any(l.split()[0] not in gx.OPS for l in body):
            skipped.append((name, [l.split()[0] for l in body if l.split()[0] not in gx.OPS]))
            continue
        for a, b, z, wv in tuples:
            w = gx.Wave(None, 0, {}, 0, "llc")
            w.s[2], w.s[3], w.s[4], w.s[5], w.s[6], w.s[7] = a, b, z & M32, z >> 32, wv & M32, wv >> 32
            for line in body:
                op, _, rest = line.partition(" ")
                if op in ("s_endpgm", "s_setpc_b64"):
                    break
                args, mods = gx._split_operands(rest.strip())
                gx.OPS[op](w, gx.Ins(op, args, mods, 0, 4, line))
                seen.add(op)
            got = (w.s[0] & M32) | ((w.s[1] & M32) << 32)
            assert got == SCALAR[name][1](a, b, z, wv) & M64, (name, hex(a), hex(b), hex(z), hex(wv), body)
    assert len(skipped) <= 3, skipped
    for op in ("s_mul_i32", "s_mul_hi_u32", "s_add_u32", "s_addc_u32", "s_sub_u32", "s_subb_u32", "s_lshl_b64", "s_lshr_b64", "s_andn2_b64", "s_orn2_b64", "s_cselect_b32", "s_bfe_u32", "s_min_u32", "s_max_i32"):
        assert op in seen, f"{op} was not exercised: {sorted(seen)}; skipped {skipped}"
