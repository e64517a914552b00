#!/usr/bin/env python3
"""Wait-state check of every inline-assembly statement of the BUILT kernels against LLVM's own gfx950 hazard recogniser (CPU only).

hipcc treats an `asm volatile` statement as one opaque instruction: it neither pads the hazards inside it nor (beyond one state)
those between it and the compiled code around it.  The kernels' asm blocks (ndzip_amd/csrc/gfx950_lds.hpp) carry hand-counted
s_nops; rounds 3-4 checked them against the builder's own reading of the ISA manual (tests/gfx950_exec.py), which missed the
gfx940-family rule "VALU writes SGPR / VCC -> VALU reads it: 2 wait states" (v_add_co -> v_addc, v_cmp -> v_cndmask).  This tool
asks the authority hipcc itself uses:

  1. `hipcc -S` both kernel translation units with the product flags;
  2. every `;;#ASMSTART ... ;;#ASMEND` block (concrete registers) is translated instruction by instruction into LLVM MIR -- only the
     few opcodes the blocks use are known, anything else is an error, so new assembly cannot slip past unchecked;
  3. each distinct block shape is wrapped in worst-case neighbours: one variant per VGPR the block reads (a VALU write of that
     register immediately in front) and one per VGPR it writes (a DPP read of it immediately behind; that also covers "VALU writes
     EXEC -> DPP: 5 states" for v_cmpx blocks);
  4. `llc -mcpu=gfx950 -run-pass post-RA-hazard-rec` runs over all of them; the check FAILS if the pass inserts a single S_NOP.
  SGPR operands of the blocks are required to be SALU-written (looked up in the listing): a VALU-written one (v_readfirstlane)
  would need 2 states in front of a VALU reader and is reported.

usage: tools/asm_hazards.py [--keep DIR]      exit status 0 = LLVM would add nothing
"""
from __future__ import annotations

import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LLC = "/opt/rocm/lib/llvm/bin/llc"

DPP_CTRL = {"row_bcast:15": 0x142, "row_bcast:31": 0x143, "wave_shl:1": 0x130, "wave_shr:1": 0x138, "row_mirror": 0x140, "row_half_mirror": 0x141}


class Unknown(Exception):
    pass


def _vreg(tok):
    m = re.fullmatch(r"v(\d+)", tok)
    if not m:
        raise Unknown(f"expected a VGPR, got {tok!r}")
    return f"$vgpr{m.group(1)}"


def _sreg(tok):
    if tok == "vcc":
        return "$vcc"
    if tok == "exec":
        return "$exec"
    m = re.fullmatch(r"s(\d+)", tok)
    if m:
        return f"$sgpr{m.group(1)}"
    m = re.fullmatch(r"s\[(\d+):(\d+)\]", tok)
    if m:
        return "$" + "_".join(f"sgpr{i}" for i in range(int(m.group(1)), int(m.group(2)) + 1))
    raise Unknown(f"expected an SGPR, got {tok!r}")


def _src(tok):
    """VOP source operand: VGPR, SGPR or integer constant."""
    if re.fullmatch(r"v\d+", tok):
        return _vreg(tok)
    if re.fullmatch(r"s\d+|s\[\d+:\d+\]|vcc|exec", tok):
        return _sreg(tok)
    return str(int(tok, 0))


def _dpp(mods):
    """-> (dpp_ctrl, row_mask, bank_mask, bound_ctrl) from the modifier text of a DPP instruction"""
    ctrl = None
    m = re.search(r"quad_perm:\[(\d),(\d),(\d),(\d)\]", mods)
    if m:
        a, b, c, d = (int(x) for x in m.groups())
        ctrl = a | b << 2 | c << 4 | d << 6
    m = re.search(r"row_(shl|shr|ror):(\d+)", mods)
    if m:
        ctrl = {"shl": 0x100, "shr": 0x110, "ror": 0x120}[m.group(1)] + int(m.group(2))
    for k, v in DPP_CTRL.items():
        if k in mods:
            ctrl = v
    if ctrl is None:
        raise Unknown(f"no DPP control in {mods!r}")
    rm = int(re.search(r"row_mask:(0x[0-9a-f]+|\d+)", mods).group(1), 0) if "row_mask" in mods else 15
    bm = int(re.search(r"bank_mask:(0x[0-9a-f]+|\d+)", mods).group(1), 0) if "bank_mask" in mods else 15
    bc = 1 if re.search(r"bound_ctrl:(1|0\b)", mods) else 0  # (the assembler spells the set bit bound_ctrl:1, older ones bound_ctrl:0)
    return ctrl, rm, bm, bc


def translate(line):
    """one line of gfx950 assembly -> (MIR text, VGPRs read, VGPRs written, SGPR tokens read by a VALU instruction)"""
    line = line.split(";")[0].strip()
    if not line:
        return None
    op, _, rest = line.partition(" ")
    rest = rest.strip()
    # split operands from trailing modifiers (DPP controls, bitop3:, offset:) -- modifiers are space separated, operands comma separated
    m = re.match(r"((?:[^,\s]+(?:\[[^\]]*\])?\s*,\s*)*[^,\s]+(?:\[[^\]]*\])?)(.*)$", rest) if rest else None
    ops = [o.strip() for o in re.split(r",\s*(?![^\[]*\])", m.group(1))] if m else []
    mods = m.group(2).strip() if m else ""
    X = "implicit $exec"
    if op == "s_nop":
        return f"S_NOP {int(ops[0], 0)}", [], [], []
    if op == "s_waitcnt":
        return "S_WAITCNT 0", [], [], []
    if op == "s_mov_b64":
        return f"{_sreg(ops[0])} = S_MOV_B64 {_sreg(ops[1])}", [], [], []
    if op in ("v_cmpx_ne_u32_e32", "v_cmpx_gt_i32_e32", "v_cmpx_lt_i32_e32", "v_cmpx_eq_u32_e32"):
        assert ops[0] == "vcc"
        return (f"{op.upper().replace('_E32', '_e32')} {_src(ops[1])}, {_vreg(ops[2])}, implicit-def $vcc, implicit-def $exec, {X}",
                [ops[2]] + ([ops[1]] if ops[1].startswith("v") else []), [], [])
    if op in ("v_cmp_ne_u32_e32", "v_cmp_eq_u32_e32", "v_cmp_gt_i32_e32", "v_cmp_lt_i32_e32"):
        assert ops[0] == "vcc"
        return (f"{op.upper().replace('_E32', '_e32')} {_src(ops[1])}, {_vreg(ops[2])}, implicit-def $vcc, {X}",
                [ops[2]] + ([ops[1]] if ops[1].startswith("v") else []), [], [])
    if op == "ds_write_b32":
        off = int(re.search(r"offset:(\d+)", mods).group(1)) if "offset" in mods else 0
        return f"DS_WRITE_B32_gfx9 {_vreg(ops[0])}, {_vreg(ops[1])}, {off}, 0, {X}", [ops[0], ops[1]], [], []
    if op in ("v_add_u32_e32", "v_lshrrev_b32_e32", "v_lshlrev_b32_e32", "v_and_b32_e32", "v_xor_b32_e32", "v_or_b32_e32"):
        reads = [o for o in ops[1:] if re.fullmatch(r"v\d+", o)]
        sreads = [o for o in ops[1:] if re.fullmatch(r"s\d+|s\[\d+:\d+\]", o)]
        return f"{_vreg(ops[0])} = {op.upper().replace('_E32', '_e32')} {_src(ops[1])}, {_vreg(ops[2])}, {X}", reads, [ops[0]], sreads
    if op == "v_bitop3_b32":
        imm = int(re.search(r"bitop3:(0x[0-9a-f]+|\d+)", mods).group(1), 0)
        reads = [o for o in ops[1:] if re.fullmatch(r"v\d+", o)]
        sreads = [o for o in ops[1:] if re.fullmatch(r"s\d+|s\[\d+:\d+\]", o)]
        return f"{_vreg(ops[0])} = V_BITOP3_B32_e64 {_src(ops[1])}, {_src(ops[2])}, {_src(ops[3])}, {imm}, {X}", reads, [ops[0]], sreads
    if op == "v_add_co_u32_dpp":
        assert ops[1] == "vcc"
        c, rm, bm, bc = _dpp(mods)
        return (f"{_vreg(ops[0])} = V_ADD_CO_U32_dpp {_vreg(ops[0])}, {_vreg(ops[2])}, {_vreg(ops[3])}, {c}, {rm}, {bm}, {bc}, implicit-def $vcc, {X}",
                [ops[0], ops[2], ops[3]], [ops[0]], [])
    if op == "v_addc_co_u32_dpp":
        assert ops[1] == "vcc" and ops[4] == "vcc"
        c, rm, bm, bc = _dpp(mods)
        return (f"{_vreg(ops[0])} = V_ADDC_U32_dpp {_vreg(ops[0])}, {_vreg(ops[2])}, {_vreg(ops[3])}, {c}, {rm}, {bm}, {bc}, implicit-def $vcc, implicit $vcc, {X}",
                [ops[0], ops[2], ops[3]], [ops[0]], [])
    if op == "v_cndmask_b32_dpp":
        assert ops[3] == "vcc"
        c, rm, bm, bc = _dpp(mods)
        return (f"{_vreg(ops[0])} = V_CNDMASK_B32_dpp undef {_vreg(ops[0])}, 0, {_vreg(ops[1])}, 0, {_vreg(ops[2])}, {c}, {rm}, {bm}, {bc}, implicit $vcc, {X}",
                [ops[1], ops[2]], [ops[0]], [])
    if op == "v_mov_b32_dpp":
        c, rm, bm, bc = _dpp(mods)
        return f"{_vreg(ops[0])} = V_MOV_B32_dpp undef {_vreg(ops[0])}, {_vreg(ops[1])}, {c}, {rm}, {bm}, {bc}, {X}", [ops[1]], [ops[0]], []
    raise Unknown(f"no MIR translation for {op!r} ({line!r}): add it to tools/asm_hazards.py")


def asm_blocks(listing):
    """[(first line number, [instruction lines], kernel name)] of every non-empty inline-asm block of a `hipcc -S` listing"""
    blocks, cur, start, kernel = [], None, 0, "?"
    for n, line in enumerate(listing, 1):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            kernel = m.group(1)
        if ";;#ASMSTART" in line:
            cur, start = [], n
        elif ";;#ASMEND" in line:
            if cur:
                blocks.append((start, cur, kernel))
            cur = None
        elif cur is not None:
            t = line.split(";")[0].strip()
            if t:
                cur.append(t)
    return blocks


def canonical(lines):
    """block text with its VGPRs / SGPRs renamed in order of appearance (hazards depend on aliasing, not on numbers)"""
    vmap, smap = {}, {}

    def v(m):
        return vmap.setdefault(m.group(0), f"v{len(vmap)}")

    def s2(m):
        key = m.group(0)
        if key not in smap:
            k = 2 * len(smap)
            smap[key] = f"s[{k + 40}:{k + 41}]"
        return smap[key]

    def s1(m):
        key = m.group(0)
        if key not in smap:
            smap[key] = f"s{2 * len(smap) + 40}"
        return smap[key]

    out = []
    for line in lines:
        line = re.sub(r"\bs\[\d+:\d+\]", s2, line)
        line = re.sub(r"\bs\d+\b", s1, line)
        line = re.sub(r"\bv\d+\b", v, line)
        out.append(line)
    return tuple(out)


def sgpr_writer_is_salu(listing, block_start, tok):
    """the last instruction in front of line `block_start` that writes SGPR `tok`: is it scalar?  (looks back 400 lines)"""
    nums = [int(x) for x in re.findall(r"\d+", tok)]
    regs = set(range(nums[0], nums[-1] + 1))
    for n in range(block_start - 2, max(0, block_start - 400), -1):
        t = listing[n].split(";")[0].strip()
        if not t or t.startswith((".", ";")) or t.endswith(":"):
            continue
        op, _, rest = t.partition(" ")
        dst = rest.split(",")[0].strip()
        m = re.fullmatch(r"s(\d+)|s\[(\d+):(\d+)\]", dst)
        if not m:
            continue
        d = set(range(int(m.group(2)), int(m.group(3)) + 1)) if m.group(2) else {int(m.group(1))}
        if d & regs:
            return op.startswith("s_"), f"line {n + 1}: {t}"
    return True, "no writer within 400 lines (kernel argument / loop-invariant scalar)"


def mir_function(name, body_lines):
    return "---\nname: %s\ntracksRegLiveness: false\nbody: |\n  bb.0:\n%s\n    S_ENDPGM 0\n...\n" % (name, "\n".join("    " + b for b in body_lines))


def check(listings, keep=None):
    """listings: {label: [lines]} -> list of problems (empty = clean); prints a summary"""
    shapes = {}  # canonical text -> (label, line, kernel)
    problems = []
    nblocks = 0
    for label, listing in listings.items():
        for start, lines, kernel in asm_blocks(listing):
            nblocks += 1
            shape = canonical(lines)
            shapes.setdefault(shape, (label, start, kernel))
            for line in lines:  # SGPRs read by VALU instructions of the block must be SALU-written
                try:
                    tr = translate(line)
                except Unknown as e:
                    problems.append(f"{label}:{start}: {e}")
                    break
                if tr:
                    for tok in tr[3]:
                        ok, why = sgpr_writer_is_salu(listing, start, tok)
                        if not ok:
                            problems.append(f"{label}:{start}: VALU instruction of the block reads {tok}, written by a VALU instruction ({why}): 2 wait states needed")
    funcs, meta = [], {}
    for i, (shape, where) in enumerate(shapes.items()):
        try:
            tr = [t for t in (translate(l) for l in shape) if t]
        except Unknown as e:
            problems.append(f"{where[0]}:{where[1]}: {e}")
            continue
        body = [t[0] for t in tr]
        written, read_first = [], []
        for _, reads, writes, _s in tr:
            for r in reads:
                if r not in written and r not in read_first:
                    read_first.append(r)
            for w in writes:
                if w not in written:
                    written.append(w)
        variants = [("plain", [], [])]
        variants += [(f"in_{r}", [f"{_vreg(r)} = V_MOV_B32_e32 0, implicit $exec"], []) for r in read_first]
        variants += [(f"out_{w}", [], [f"$vgpr255 = V_MOV_B32_dpp undef $vgpr255, {_vreg(w)}, 177, 15, 15, -1, implicit $exec"]) for w in written]
        if any("V_CMPX" in b for b in body) and not written:
            variants.append(("out_exec", [], ["$vgpr255 = V_MOV_B32_dpp undef $vgpr255, $vgpr254, 177, 15, 15, -1, implicit $exec"]))
        for vname, pro, epi in variants:
            name = f"b{i}_{vname}"
            funcs.append(mir_function(name, pro + body + epi))
            meta[name] = (where, sum(1 for b in pro + body + epi if b.startswith("S_NOP")), shape)
    with tempfile.TemporaryDirectory() as tmp:
        d = keep or tmp
        os.makedirs(d, exist_ok=True)
        src = os.path.join(d, "asm_blocks.mir")
        with open(src, "w") as f:
            f.write("".join(funcs))
        r = subprocess.run([LLC, "-mtriple=amdgcn-amd-amdhsa", "-mcpu=gfx950", "-run-pass", "post-RA-hazard-rec", "-o", "-", src], capture_output=True, text=True)
        if r.returncode != 0:
            problems.append("llc failed on the translated blocks:\n" + r.stderr[-2000:])
            return problems
        if keep:
            with open(os.path.join(d, "asm_blocks.out.mir"), "w") as f:
                f.write(r.stdout)
    name, nops, prev = None, 0, None
    seen = set()

    def close():
        if name is not None and nops != meta[name][1]:
            where, before, shape = meta[name]
            problems.append(f"{where[0]}:{where[1]} ({where[2][:60]}), variant {name}: LLVM's gfx950 hazard recogniser inserts {nops - before} more "
                            f"S_NOP instruction(s) -- first one behind `{first_after}`\n      block: " + " | ".join(shape[:6]) + (" ..." if len(shape) > 6 else ""))

    first_after = None
    for line in r.stdout.splitlines():
        m = re.match(r"name:\s+(\S+)", line)
        if m:
            close()
            name, nops, prev, first_after = m.group(1), 0, None, None
            seen.add(name)
            continue
        t = line.strip()
        if name and re.match(r"(\$|S_|V_|DS_)", t):
            if t.startswith("S_NOP"):
                nops += 1
                if nops > meta[name][1] and first_after is None:
                    first_after = prev
            prev = t
    close()
    missing = set(meta) - seen
    if missing:
        problems.append(f"llc dropped {len(missing)} function(s): {sorted(missing)[:3]}")
    print(f"asm_hazards: {nblocks} inline-asm blocks, {len(shapes)} distinct shapes, {len(funcs)} MIR variants through post-RA-hazard-rec: "
          f"{'clean' if not problems else str(len(problems)) + ' problem(s)'}")
    return problems


def listings_of_build(extra_flags=()):
    from ndzip_amd import build

    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        for src in ("kernels_f32.hip", "kernels_f64.hip"):
            s = os.path.join(tmp, src.replace(".hip", ".s"))
            subprocess.run([build.HIPCC, *build.FLAGS, *extra_flags, "-S", "--cuda-device-only", "-o", s, os.path.join(build.CSRC, src)], check=True, capture_output=True)
            out[src] = open(s).read().splitlines()
    return out


if __name__ == "__main__":
    keep = sys.argv[sys.argv.index("--keep") + 1] if "--keep" in sys.argv else None
    probs = check(listings_of_build(), keep=keep)
    for p in probs:
        print("  " + p)
    sys.exit(1 if probs else 0)
