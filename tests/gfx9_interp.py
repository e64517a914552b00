"""TEST INFRASTRUCTURE ONLY: an interpreter for the handful of gfx9 instructions the product's hand-written EXEC-masked assembly
consists of (ndzip_amd/csrc/gfx950_lds.hpp: lds_append_nonzero), so that the CPU suite EXECUTES the sequence that is in the built
code object -- as disassembled by llvm-objdump -- for 64 lanes, instead of only checking its shape.  Semantics follow the
"Vega" / CDNA3 ISA manuals:

  s_mov_b64 sD, sS        SGPR pair / EXEC / VCC copy
  v_cmpx_ne_u32_e32 vcc, S0, V1   per ACTIVE lane r = S0 != V1; VCC = r (0 in inactive lanes); EXEC = r   (gfx9: v_cmpx writes both)
  ds_write_b32 vA, vD [offset:N]  per active lane LDS[vA + N] = vD (dword)
  v_add_u32_e32 vD, S0, V1        per active lane vD = S0 + V1 (mod 2^32); inactive lanes keep vD
  v_mov_b32_e32 vD, S0            per active lane
  s_nop N                         nothing

Anything else raises: the point is that NOTHING else may sit inside a masked stretch."""
from __future__ import annotations

import re

import numpy as np

MASK64 = (1 << 64) - 1


class Unsupported(Exception):
    pass


class Wave:
    def __init__(self, lds_bytes: int = 64 * 1024):
        self.v: dict[str, np.ndarray] = {}
        self.s: dict[str, int] = {}     # "s[8:9]" -> 64-bit value, "s4" -> 32-bit
        self.exec = MASK64
        self.vcc = 0
        self.lds = np.zeros(lds_bytes // 4, dtype=np.uint32)
        self.lds_written = np.zeros(lds_bytes // 4, dtype=bool)
        self.trace: list[str] = []

    def active(self) -> np.ndarray:
        return np.array([(self.exec >> i) & 1 for i in range(64)], dtype=bool)

    def vreg(self, name: str) -> np.ndarray:
        if name not in self.v:
            raise Unsupported(f"read of {name}, which nothing has set")
        return self.v[name]

    def src(self, tok: str) -> np.ndarray:
        tok = tok.strip()
        if re.fullmatch(r"v\d+", tok):
            return self.vreg(tok)
        if re.fullmatch(r"-?\d+", tok):
            return np.full(64, int(tok) & 0xFFFFFFFF, dtype=np.uint32)
        if re.fullmatch(r"0x[0-9a-fA-F]+", tok):
            return np.full(64, int(tok, 16) & 0xFFFFFFFF, dtype=np.uint32)
        if re.fullmatch(r"s\d+", tok):
            return np.full(64, self.s[tok] & 0xFFFFFFFF, dtype=np.uint32)
        raise Unsupported(f"operand {tok!r}")

    def read64(self, tok: str) -> int:
        tok = tok.strip()
        if tok == "exec":
            return self.exec
        if tok == "vcc":
            return self.vcc
        if re.fullmatch(r"s\[\d+:\d+\]", tok):
            if tok not in self.s:
                raise Unsupported(f"read of {tok}, which nothing has set")
            return self.s[tok]
        raise Unsupported(f"64-bit operand {tok!r}")

    def write64(self, tok: str, value: int):
        tok = tok.strip()
        if tok == "exec":
            self.exec = value & MASK64
        elif tok == "vcc":
            self.vcc = value & MASK64
        elif re.fullmatch(r"s\[\d+:\d+\]", tok):
            self.s[tok] = value & MASK64
        else:
            raise Unsupported(f"64-bit destination {tok!r}")

    def step(self, ins: str):
        self.trace.append(ins)
        op, _, rest = ins.partition(" ")
        args = [a.strip() for a in rest.split(",")] if rest else []
        if op == "s_nop":
            return
        if op == "s_mov_b64":
            self.write64(args[0], self.read64(args[1]))
            return
        if op == "v_cmpx_ne_u32_e32":
            if args[0] != "vcc":
                raise Unsupported(ins)
            a, b = self.src(args[1]), self.src(args[2])
            act = self.active()
            r = (a != b) & act
            bits = sum(1 << i for i in range(64) if r[i])
            self.vcc = bits
            self.exec = bits
            return
        if op == "ds_write_b32":
            m = re.fullmatch(r"(v\d+)(?: offset:(\d+))?", args[1]) if len(args) > 1 else None
            if not m:
                raise Unsupported(ins)
            addr = self.vreg(args[0]).astype(np.uint64) + (int(m.group(2)) if m.group(2) else 0)
            data = self.vreg(m.group(1))
            for lane in np.nonzero(self.active())[0]:
                a = int(addr[lane])
                if a % 4 or a // 4 >= self.lds.size:
                    raise Unsupported(f"ds_write_b32 to LDS byte address {a}")
                self.lds[a // 4] = data[lane]
                self.lds_written[a // 4] = True
            return
        if op in ("v_add_u32_e32", "v_mov_b32_e32"):
            if not re.fullmatch(r"v\d+", args[0]):
                raise Unsupported(ins)
            if op == "v_add_u32_e32":
                val = (self.src(args[1]).astype(np.uint64) + self.src(args[2]).astype(np.uint64)).astype(np.uint32)
            else:
                val = self.src(args[1]).copy()
            old = self.v.get(args[0], np.zeros(64, dtype=np.uint32))
            self.v[args[0]] = np.where(self.active(), val, old).astype(np.uint32)
            return
        raise Unsupported(ins)


def disassemble(lib_path: str, workdir: str) -> list[str]:
    """instruction texts of every gfx950 code object bundled in the library, in order"""
    import os
    import shutil
    import subprocess

    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    lib = shutil.copy(lib_path, os.path.join(workdir, "lib.so"))
    subprocess.run([objdump, "--offloading", str(lib)], cwd=workdir, capture_output=True, text=True, check=True)
    text = ""
    for f in sorted(os.listdir(workdir)):
        if "gfx950" in f:
            text += subprocess.run([objdump, "-d", os.path.join(workdir, f)], capture_output=True, text=True, check=True).stdout
    return [line.split("\t")[1].split("//")[0].strip() for line in text.splitlines()
            if line.startswith("\t") and len(line.split("\t")) > 1 and line.split("\t")[1].strip()]
