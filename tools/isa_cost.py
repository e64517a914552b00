#!/usr/bin/env python
"""Static per-iteration instruction budget of a persistent kernel's main loop, from a `hipcc -S` file (CPU only).

usage: isa_cost.py file.s <substring of the mangled kernel name>

Blocks are attributed by LLVM's own loop comments (`; in Loop: Header=BBn_m Depth=d`): everything at depth 1 of the outermost
loop is executed (at most) once per iteration; deeper blocks (the copy-out loops, the look-back's retry loops) are listed
separately with their static size.  Issue cost per wave-instruction (MI355X guide: a wave64 VALU instruction occupies its SIMD
for 4 cycles; LDS per the guide's table; SALU, branches and waits issue on their own ports, 1 cycle here):
    VALU 4 | ds_read_b128 4, ds_read_b64 2, ds_read_b32 2, ds_write_b32 4, ds_write_b64 6, ds_write_b128 13, ds_bpermute 4
    | VMEM 4 | SALU / branch / waitcnt / nop 1 (s_nop N: N + 1)
This is an ISSUE budget -- a lower bound per wavefront that ignores every stall -- good for A/B-ing code-generation changes
offline, not a prediction of run time."""
import collections
import re
import sys

LDS = {"ds_read_b128": 4, "ds_read_b64": 2, "ds_read_b32": 2, "ds_read2_b64": 8, "ds_read2_b32": 4, "ds_write_b32": 4, "ds_write_b64": 6,
       "ds_write_b128": 13, "ds_write2_b64": 13, "ds_bpermute_b32": 4}


def cost(op, args):
    if op.startswith("v_"):
        return "valu", 4
    if op.startswith("ds_"):
        return "lds", LDS.get(op, 4)
    if op.startswith(("global_", "buffer_", "flat_")):
        return "vmem", 4
    if op == "s_nop":
        return "salu", int(args.split()[0]) + 1 if args else 1
    return "salu", 1


def main():
    text = open(sys.argv[1]).read()
    key = sys.argv[2]
    for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)\n\s*s_endpgm", text, re.S | re.M):
        name, body = m.group(1), m.group(2)
        if key not in name:
            continue
        depth, header = 0, None
        per = collections.defaultdict(lambda: collections.Counter())
        ops = collections.defaultdict(lambda: collections.Counter())
        outer = None
        for line in body.split("\n"):
            t = line.strip()
            mm = re.search(r";\s+(?:in Loop: Header=|=>This (?:Inner )?Loop Header: Depth=|Parent Loop )", t)
            lab = re.match(r"^(\.LBB\d+_\d+):", t) or re.match(r"^; %bb\.\d+:", t)
            if lab:
                d = re.search(r"Depth[= ](\d+)", t)
                inl = re.search(r"in Loop: Header=(BB\d+_\d+) Depth=(\d+)", t)
                hdr = re.search(r"This (?:Inner )?Loop Header: Depth=(\d+)", t)
                if inl:
                    depth = int(inl.group(2))
                elif hdr:
                    depth = int(hdr.group(1))
                else:
                    depth = 0
                continue
            if not t or t[0] in ".;/" or t.endswith(":"):
                continue
            parts = t.split(None, 1)
            op, args = parts[0], parts[1] if len(parts) > 1 else ""
            cls, c = cost(op, args)
            per[depth][cls] += c
            per[depth]["n_" + cls] += 1
            ops[depth][op] += 1
        print(name[:100])
        for d in sorted(per):
            p = per[d]
            where = {0: "outside loops (prologue / epilogue / out-of-line)", 1: "main loop, once per iteration"}.get(d, f"inner loops (depth {d}), per trip")
            print(f"  depth {d} [{where}]: VALU {p['n_valu']} instr = {p['valu']} cyc | LDS {p['n_lds']} = {p['lds']} cyc | VMEM {p['n_vmem']} | "
                  f"SALU-side {p['n_salu']} = {p['salu']} cyc")
        top = ", ".join(f"{k}:{v}" for k, v in ops[1].most_common(18))
        print("  main loop ops:", top)


if __name__ == "__main__":
    main()
