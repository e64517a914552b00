"""TEST INFRASTRUCTURE ONLY: executes kernels of the BUILT gfx950 code objects (the ones inside ndzip_amd/libndzip_hip.so and
libndzip_hip_stages.so) on the CPU, instruction by instruction, 64 lanes per wavefront -- so that, in a container without a GPU,
what is checked against the oracle is not only the kernels' C++ logic (tests/wavesim compiles the sources for the host) but the
code hipcc actually generated: register allocation around the inline assembly, the EXEC-masked plane compaction, every DPP control
word, the v_readfirstlane uniformity pins, scalar-base addressing, s_waitcnt-free semantics of loads and stores, the ticket /
look-back protocol between concurrently running workgroups.

How it is driven: the functional model (tests/wavesim) runs the unchanged HOST side of the library (capi.hip: argument checks,
geometry, launch sequences, scratch management); its kernel launches are intercepted (wavesim_set_launch_hook) and handed to this
interpreter, which finds the kernel of the same mangled name in the gfx950 code object, lays the kernel arguments out as the HSA ABI
says (explicit arguments as packed by the model, hidden arguments from the code object's metadata) and runs the grid.  "Device
memory" is the process's own memory (numpy buffers of the tests, the model's hipMalloc = malloc).

What it is NOT: a timing model or a memory-model checker (one sequentially consistent memory; s_nop / s_waitcnt change no result:
every instruction completes before the next starts).  It does keep wait-state and waitcnt BOOKKEEPING on the executed stream
(check_hazards / check_waits below) -- a hazard table written from the builder's reading of the ISA manual, which missed the
gfx940-family "VALU writes SGPR / VCC -> VALU reads it: 2 wait states" until round 5; the authority for the inline assembly's wait
states is tools/asm_hazards.py (LLVM's own hazard recogniser), this table is a second net under it.  Semantics follow the Vega / CDNA3 ISA manuals as far as the
instructions that occur in these kernels go; anything else raises Unsupported with the instruction text.
Not part of the product: nothing under ndzip_amd/ imports this."""
from __future__ import annotations

import ctypes as C
import os
import random
import re
import struct
import subprocess

import numpy as np

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
M32 = 0xFFFFFFFF
M64 = (1 << 64) - 1
LANES = np.arange(64, dtype=np.int64)


class Unsupported(Exception):
    pass


# ------------------------------------------------------------------------------------------------------------------------------
# code objects
# ------------------------------------------------------------------------------------------------------------------------------

class Ins:
    __slots__ = ("op", "args", "mods", "addr", "size", "text", "fn", "target")

    def __init__(self, op, args, mods, addr, size, text):
        self.op, self.args, self.mods, self.addr, self.size, self.text = op, args, mods, addr, size, text
        self.fn = None
        self.target = None


def _split_operands(rest: str):
    """'v[4:7], v8, s[2:3] offset:16 nt' -> (['v[4:7]', 'v8', 's[2:3]'], {'offset': '16', 'nt': True})"""
    parts, depth, cur = [], 0, ""
    for ch in rest:
        if ch in "[(":
            depth += 1
        elif ch in "])":
            depth -= 1
        if ch == "," and depth == 0:
            parts.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        parts.append(cur.strip())
    mods = {}
    if parts:
        toks = parts[-1].split()
        # the last operand may be followed by modifiers; an instruction without operands (s_waitcnt vmcnt(0)) has only "modifiers"
        keep = []
        for i, t in enumerate(toks):
            if i == 0 and not re.match(r"^(offset\d?|row_\w+|quad_perm|bank_mask|bound_ctrl|bitop3|op_sel\w*|vmcnt|lgkmcnt|expcnt|nt|sc0|sc1|glc|slc|clamp|gds)(:|\(|$)", t):
                keep.append(t)
            else:
                if ":" in t:
                    k, v = t.split(":", 1)
                    mods[k] = v
                elif "(" in t:
                    k, v = t.split("(", 1)
                    mods[k] = v.rstrip(")")
                else:
                    mods[t] = True
        if keep:
            parts[-1] = keep[0]
        else:
            parts.pop()
    return parts, mods


class CodeObject:
    """One gfx950 ELF: kernels by name, their descriptors and metadata, the text as parsed instructions."""

    def __init__(self, path: str):
        self.path = path
        data = open(path, "rb").read()
        self.data = data
        assert data[:4] == b"\x7fELF" and data[4] == 2 and data[5] == 1, "ELF64 little-endian expected"
        shoff, = struct.unpack_from("<Q", data, 0x28)
        shentsize, shnum, shstrndx = struct.unpack_from("<HHH", data, 0x3A)
        secs = []
        for i in range(shnum):
            name, typ, flags, addr, off, size, link, info, align, entsize = struct.unpack_from("<IIQQQQIIQQ", data, shoff + i * shentsize)
            secs.append(dict(name=name, type=typ, addr=addr, off=off, size=size, link=link, entsize=entsize))
        strtab = secs[shstrndx]

        def sname(o, tab=strtab):
            end = data.index(b"\0", tab["off"] + o)
            return data[tab["off"] + o:end].decode()

        for s in secs:
            s["name"] = sname(s["name"])
        self.sections = {s["name"]: s for s in secs}
        self.symbols = {}
        symtab = self.sections.get(".symtab")
        if symtab:
            st = secs[symtab["link"]]
            for i in range(symtab["size"] // 24):
                nm, info, other, shndx, value, size = struct.unpack_from("<IBBHQQ", data, symtab["off"] + 24 * i)
                n = sname(nm, st)
                if n:
                    self.symbols[n] = (value, size)
        # metadata (MessagePack in the NT_AMDGPU_METADATA note)
        import msgpack

        note = self.sections[".note"]
        p, end = note["off"], note["off"] + note["size"]
        self.meta = None
        while p < end:
            namesz, descsz, typ = struct.unpack_from("<III", data, p)
            p += 12
            p += (namesz + 3) & ~3
            desc = data[p:p + descsz]
            p += (descsz + 3) & ~3
            if typ == 32:
                self.meta = msgpack.unpackb(desc, raw=False, strict_map_key=False)
        self.kernels = {k[".name"]: k for k in self.meta["amdhsa.kernels"]}
        self._ins = None

    def read_vaddr(self, vaddr: int, n: int) -> bytes:
        for s in self.sections.values():
            if s["addr"] <= vaddr < s["addr"] + s["size"] and s["type"] != 8:
                o = s["off"] + (vaddr - s["addr"])
                return self.data[o:o + n]
        raise KeyError(hex(vaddr))

    def descriptor(self, name: str) -> dict:
        kd_addr, _ = self.symbols[name + ".kd"]
        kd = self.read_vaddr(kd_addr, 64)
        entry_off, = struct.unpack_from("<q", kd, 16)
        rsrc1, rsrc2, props, preload = struct.unpack_from("<IIHH", kd, 48)
        return dict(entry=kd_addr + entry_off, rsrc1=rsrc1, rsrc2=rsrc2, props=props, preload=preload,
                    user_sgprs=(rsrc2 >> 1) & 31, wg_id_x=(rsrc2 >> 7) & 1, wg_id_y=(rsrc2 >> 8) & 1, wg_id_z=(rsrc2 >> 9) & 1,
                    wg_info=(rsrc2 >> 10) & 1, vgpr_workitem_id=(rsrc2 >> 11) & 3, private_segment=rsrc2 & 1,
                    lds_fixed=struct.unpack_from("<I", kd, 0)[0])

    def instructions(self):
        """{address: Ins}, in address order (a dict keeps insertion order), over the whole .text"""
        if self._ins is None:
            out = subprocess.run([OBJDUMP, "-d", self.path], capture_output=True, text=True, check=True).stdout
            ins = {}
            for line in out.splitlines():
                if not line.startswith("\t"):
                    continue
                body, _, comment = line.partition("//")
                m = re.match(r"\s*([0-9A-Fa-f]+):\s*((?:[0-9A-Fa-f]{8}\s*)+)", comment)
                if not m:
                    continue
                addr = int(m.group(1), 16)
                size = 4 * len(m.group(2).split())
                text = body.strip()
                op, _, rest = text.partition(" ")
                args, mods = _split_operands(rest.strip())
                ins[addr] = Ins(op, args, mods, addr, size, text)
            self._ins = ins
        return self._ins


def code_objects_of(lib_path: str, workdir: str):
    """the gfx950 code objects bundled in a HIP shared library"""
    import shutil

    os.makedirs(workdir, exist_ok=True)
    lib = shutil.copy(lib_path, os.path.join(workdir, "lib.so"))
    subprocess.run([OBJDUMP, "--offloading", lib], cwd=workdir, capture_output=True, text=True, check=True)
    return [CodeObject(os.path.join(workdir, f)) for f in sorted(os.listdir(workdir)) if "gfx950" in f]


# ------------------------------------------------------------------------------------------------------------------------------
# process memory as "device memory"
# ------------------------------------------------------------------------------------------------------------------------------

def _span(lo: int, n: int) -> np.ndarray:
    return np.frombuffer((C.c_uint8 * n).from_address(lo), dtype=np.uint8)


def mem_gather(addrs: np.ndarray, active: np.ndarray, nbytes: int) -> np.ndarray:
    """bytes [lane, nbytes] at the per-lane addresses (inactive lanes: zeros)"""
    out = np.zeros((64, nbytes), dtype=np.uint8)
    if not active.any():
        return out
    a = addrs[active].astype(np.uint64)
    lo, hi = int(a.min()), int(a.max()) + nbytes
    if hi - lo > (1 << 31):
        raise Unsupported(f"one vector memory instruction spans {hi - lo} bytes: lanes in different allocations?")
    span = _span(lo, hi - lo)
    idx = (a - np.uint64(lo)).astype(np.int64)[:, None] + np.arange(nbytes, dtype=np.int64)[None, :]
    out[active] = span[idx]
    return out


def mem_scatter(addrs: np.ndarray, active: np.ndarray, data: np.ndarray):
    """data [lane, nbytes] uint8 to the per-lane addresses; lanes in ascending order (a later lane wins on overlap)"""
    if not active.any():
        return
    nbytes = data.shape[1]
    a = addrs[active].astype(np.uint64)
    lo, hi = int(a.min()), int(a.max()) + nbytes
    if hi - lo > (1 << 31):
        raise Unsupported(f"one vector memory instruction spans {hi - lo} bytes: lanes in different allocations?")
    span = _span(lo, hi - lo)
    idx = (a - np.uint64(lo)).astype(np.int64)[:, None] + np.arange(nbytes, dtype=np.int64)[None, :]
    span[idx] = data[active]


# ------------------------------------------------------------------------------------------------------------------------------
# one wavefront
# ------------------------------------------------------------------------------------------------------------------------------

_FLOAT_CONST = {"0.5": 0x3F000000, "-0.5": 0xBF000000, "1.0": 0x3F800000, "-1.0": 0xBF800000, "2.0": 0x40000000, "-2.0": 0xC0000000,
                "4.0": 0x40800000, "-4.0": 0xC0800000}
_RE_V = re.compile(r"^v(\d+)$")
_RE_VR = re.compile(r"^v\[(\d+):(\d+)\]$")
_RE_S = re.compile(r"^s(\d+)$")
_RE_SR = re.compile(r"^s\[(\d+):(\d+)\]$")
_MASKS = {}


def mask_of(bits: int) -> np.ndarray:
    m = _MASKS.get(bits)
    if m is None:
        m = np.array([(bits >> i) & 1 for i in range(64)], dtype=bool)
        if len(_MASKS) < 4096:
            _MASKS[bits] = m
    return m


def bits_of(mask: np.ndarray) -> int:
    return int(np.packbits(mask.astype(np.uint8), bitorder="little").view(np.uint64)[0])


def _imm(tok: str):
    if tok in _FLOAT_CONST:
        return _FLOAT_CONST[tok]
    try:
        return int(tok, 0)
    except ValueError:
        return None


class Workgroup:
    def __init__(self, index: int, nwaves: int, lds_bytes: int):
        self.index = index
        self.lds = np.zeros(((lds_bytes + 15) // 16 + 1) * 16, dtype=np.uint8)
        self.lds[:] = 0xA5  # (LDS is not zeroed on hardware: junk, like the functional model)
        self.waves = []
        self.at_barrier = 0
        self.live = nwaves


class Wave:
    RUNNING, BARRIER, YIELD, DONE = range(4)

    def __init__(self, wg: Workgroup, index: int, code: dict, entry: int, kernel_name: str = ""):
        self.wg, self.index, self.code, self.kernel_name = wg, index, code, kernel_name
        self.pc = entry
        self.s = [0xBAD0BAD0] * 128  # (registers hold junk at launch: whatever the previous wavefront left)
        self.v = np.zeros((512, 64), dtype=np.uint32)
        self.v[:] = 0xDEADBEEF
        self.exec = M64
        self.vcc = 0xBAD0BAD0BAD0BAD0
        self.scc = 1
        self.m0 = 0xBAD0BAD0
        self.state = Wave.RUNNING
        self.count = 0
        # hazard bookkeeping (check_hazards): wait-state clock, and when VALU last wrote each VGPR / SGPR / EXEC
        self.clock = 0
        self.vgpr_written = {}
        self.sgpr_written = {}
        self.exec_written = -100
        # s_waitcnt bookkeeping (check_waits): memory operations issued and not yet waited for, oldest first
        self.vm_queue = []    # [set of destination VGPRs] per vector-memory operation (empty set: a store)
        self.lgkm_queue = []  # [(kind, set of destination registers)]: kind "lds" (VGPRs), "ldsw" (an LDS store), "smem" (SGPRs, as -1 - n)
        self.pending = {}     # register -> number of queued operations that will write it (VGPR n: n, SGPR n: -1 - n)

    # ---- operand access ---------------------------------------------------------------------------------------------------
    def rs32(self, tok: str) -> int:
        m = _RE_S.match(tok)
        if m:
            return self.s[int(m.group(1))]
        v = _imm(tok)
        if v is not None:
            return v & M32
        if tok == "vcc_lo":
            return self.vcc & M32
        if tok == "vcc_hi":
            return self.vcc >> 32
        if tok == "exec_lo":
            return self.exec & M32
        if tok == "exec_hi":
            return self.exec >> 32
        if tok == "m0":
            return self.m0
        if tok == "scc":
            return self.scc
        if tok == "vcc":  # (a 64-bit register named where 32 bits are read: the low half)
            return self.vcc & M32
        if tok == "exec":
            return self.exec & M32
        raise Unsupported(f"scalar operand {tok!r}")

    def ws32(self, tok: str, val: int):
        val &= M32
        m = _RE_S.match(tok)
        if m:
            self.s[int(m.group(1))] = val
        elif tok == "vcc_lo":
            self.vcc = (self.vcc & ~M32) | val
        elif tok == "vcc_hi":
            self.vcc = (self.vcc & M32) | (val << 32)
        elif tok == "exec_lo":
            self.exec = (self.exec & ~M32) | val
        elif tok == "exec_hi":
            self.exec = (self.exec & M32) | (val << 32)
        elif tok == "m0":
            self.m0 = val
        else:
            raise Unsupported(f"scalar destination {tok!r}")

    def rs64(self, tok: str) -> int:
        m = _RE_SR.match(tok)
        if m:
            a = int(m.group(1))
            return self.s[a] | (self.s[a + 1] << 32)
        if tok == "vcc":
            return self.vcc
        if tok == "exec":
            return self.exec
        if tok in _FLOAT_CONST:
            # (a float inline constant in a 64-bit operand is the DOUBLE of that value: LLVM writes 1 << 62 as `v_mov_b64 v[a:b], 2.0`;
            # found by tools/fuzz_interpreter_vs_compiler.py -- the product's code objects hold no such operand)
            return struct.unpack("<Q", struct.pack("<d", float(tok)))[0]
        v = _imm(tok)
        if v is not None:
            return v & M64  # (inline integer constants are sign-extended to 64 bits; a 32-bit literal is zero-extended)
        raise Unsupported(f"64-bit scalar operand {tok!r}")

    def ws64(self, tok: str, val: int):
        val &= M64
        m = _RE_SR.match(tok)
        if m:
            a = int(m.group(1))
            self.s[a], self.s[a + 1] = val & M32, val >> 32
        elif tok == "vcc":
            self.vcc = val
        elif tok == "exec":
            self.exec = val
        else:
            raise Unsupported(f"64-bit scalar destination {tok!r}")

    def rv32(self, tok: str):
        """vector source: a [64] uint32 array, or a scalar broadcast as np.uint32"""
        m = _RE_V.match(tok)
        if m:
            return self.v[int(m.group(1))]
        return np.uint32(self.rs32(tok))

    def rv64(self, tok: str):
        m = _RE_VR.match(tok)
        if m:
            a = int(m.group(1))
            return self.v[a].astype(np.uint64) | (self.v[a + 1].astype(np.uint64) << np.uint64(32))
        return np.uint64(self.rs64(tok))

    def wv32(self, tok: str, val, mask=None):
        m = _RE_V.match(tok)
        if not m:
            raise Unsupported(f"vector destination {tok!r}")
        n = int(m.group(1))
        if mask is None:
            mask = mask_of(self.exec)
        self.v[n] = np.where(mask, np.asarray(val).astype(np.uint32), self.v[n])

    def wv64(self, tok: str, val):
        m = _RE_VR.match(tok)
        if not m:
            raise Unsupported(f"64-bit vector destination {tok!r}")
        n = int(m.group(1))
        mask = mask_of(self.exec)
        val = np.asarray(val).astype(np.uint64)
        self.v[n] = np.where(mask, (val & np.uint64(M32)).astype(np.uint32), self.v[n])
        self.v[n + 1] = np.where(mask, (val >> np.uint64(32)).astype(np.uint32), self.v[n + 1])

    def vrange(self, tok: str):
        m = _RE_VR.match(tok)
        if m:
            return int(m.group(1)), int(m.group(2)) - int(m.group(1)) + 1
        m = _RE_V.match(tok)
        if m:
            return int(m.group(1)), 1
        raise Unsupported(f"vector register (range) {tok!r}")

    def wmask(self, tok: str, mask: np.ndarray):
        """lane mask result of a compare: inactive lanes read as 0"""
        bits = bits_of(mask & mask_of(self.exec))
        self.ws64(tok, bits)


def _i32(x):
    return np.asarray(x).astype(np.uint32).view(np.int32) if isinstance(x, np.ndarray) else np.int32(np.uint32(x))


def _sx(v: int, bits: int) -> int:
    v &= (1 << bits) - 1
    return v - (1 << bits) if v >> (bits - 1) else v


# ------------------------------------------------------------------------------------------------------------------------------
# instruction semantics
# ------------------------------------------------------------------------------------------------------------------------------

OPS = {}


def op(*names):
    def deco(fn):
        for n in names:
            OPS[n] = fn
        return fn
    return deco


def _u32(x):
    return np.asarray(x, dtype=np.uint64) & np.uint64(M32)


# ---- SALU -----------------------------------------------------------------------------------------------------------------------

@op("s_nop", "s_waitcnt", "s_setprio", "s_inst_prefetch", "s_code_end")
def _(w, i):
    pass


@op("s_sleep")
def _(w, i):
    w.state = Wave.YIELD


@op("s_barrier")
def _(w, i):
    w.state = Wave.BARRIER


@op("s_endpgm")
def _(w, i):
    w.state = Wave.DONE


@op("s_mov_b32")
def _(w, i):
    w.ws32(i.args[0], w.rs32(i.args[1]))


@op("s_mov_b64")
def _(w, i):
    w.ws64(i.args[0], w.rs64(i.args[1]))


@op("s_movk_i32")
def _(w, i):
    w.ws32(i.args[0], _sx(int(i.args[1], 0), 16))


@op("s_brev_b32")
def _(w, i):
    w.ws32(i.args[0], int(f"{w.rs32(i.args[1]):032b}"[::-1], 2))


@op("s_ff1_i32_b64")
def _(w, i):
    v = w.rs64(i.args[1])
    w.ws32(i.args[0], (v & -v).bit_length() - 1 if v else M32)


@op("s_ff1_i32_b32")
def _(w, i):
    v = w.rs32(i.args[1])
    w.ws32(i.args[0], (v & -v).bit_length() - 1 if v else M32)


@op("s_bcnt1_i32_b64")
def _(w, i):
    r = bin(w.rs64(i.args[1])).count("1")
    w.ws32(i.args[0], r)
    w.scc = int(r != 0)


@op("s_not_b32")
def _(w, i):
    r = ~w.rs32(i.args[1]) & M32
    w.ws32(i.args[0], r)
    w.scc = int(r != 0)


@op("s_not_b64")
def _(w, i):
    r = ~w.rs64(i.args[1]) & M64
    w.ws64(i.args[0], r)
    w.scc = int(r != 0)


def _salu2(name, bits, fn):
    rd = Wave.rs32 if bits == 32 else Wave.rs64
    wr = Wave.ws32 if bits == 32 else Wave.ws64
    mask = M32 if bits == 32 else M64

    def h(w, i):
        r, scc = fn(rd(w, i.args[1]), rd(w, i.args[2]), w.scc)
        wr(w, i.args[0], r & mask)
        if scc is not None:
            w.scc = int(scc)
    OPS[name] = h


def _ovf_add(a, b):
    r = _sx(a, 32) + _sx(b, 32)
    return r & M32, not (-(1 << 31) <= r < (1 << 31))


def _ovf_sub(a, b):
    r = _sx(a, 32) - _sx(b, 32)
    return r & M32, not (-(1 << 31) <= r < (1 << 31))


_salu2("s_add_u32", 32, lambda a, b, c: ((a + b) & M32, (a + b) >> 32))
_salu2("s_addc_u32", 32, lambda a, b, c: ((a + b + c) & M32, (a + b + c) >> 32))
_salu2("s_sub_u32", 32, lambda a, b, c: ((a - b) & M32, b > a))
_salu2("s_subb_u32", 32, lambda a, b, c: ((a - b - c) & M32, b + c > a))
_salu2("s_add_i32", 32, lambda a, b, c: _ovf_add(a, b))
_salu2("s_sub_i32", 32, lambda a, b, c: _ovf_sub(a, b))
_salu2("s_mul_i32", 32, lambda a, b, c: ((a * b) & M32, None))
_salu2("s_mul_hi_u32", 32, lambda a, b, c: ((a * b) >> 32, None))
_salu2("s_mul_hi_i32", 32, lambda a, b, c: (((_sx(a, 32) * _sx(b, 32)) >> 32) & M32, None))
_salu2("s_min_u32", 32, lambda a, b, c: (min(a, b), a < b))
_salu2("s_max_u32", 32, lambda a, b, c: (max(a, b), a > b))
_salu2("s_min_i32", 32, lambda a, b, c: ((a if _sx(a, 32) < _sx(b, 32) else b), _sx(a, 32) < _sx(b, 32)))
_salu2("s_max_i32", 32, lambda a, b, c: ((a if _sx(a, 32) > _sx(b, 32) else b), _sx(a, 32) > _sx(b, 32)))
_salu2("s_lshl_b32", 32, lambda a, b, c: ((a << (b & 31)) & M32, ((a << (b & 31)) & M32) != 0))
_salu2("s_lshr_b32", 32, lambda a, b, c: (a >> (b & 31), (a >> (b & 31)) != 0))
_salu2("s_ashr_i32", 32, lambda a, b, c: ((_sx(a, 32) >> (b & 31)) & M32, ((_sx(a, 32) >> (b & 31)) & M32) != 0))
_salu2("s_bfe_u32", 32, lambda a, b, c: ((a >> (b & 31)) & ((1 << ((b >> 16) & 127)) - 1), ((a >> (b & 31)) & ((1 << ((b >> 16) & 127)) - 1)) != 0))
for _n, _f in (("and", lambda a, b: a & b), ("or", lambda a, b: a | b), ("xor", lambda a, b: a ^ b), ("andn2", lambda a, b: a & ~b),
               ("orn2", lambda a, b: a | ~b), ("nand", lambda a, b: ~(a & b)), ("nor", lambda a, b: ~(a | b)), ("xnor", lambda a, b: ~(a ^ b))):
    _salu2(f"s_{_n}_b32", 32, (lambda f: lambda a, b, c: (f(a, b) & M32, (f(a, b) & M32) != 0))(_f))
    _salu2(f"s_{_n}_b64", 64, (lambda f: lambda a, b, c: (f(a, b) & M64, (f(a, b) & M64) != 0))(_f))


@op("s_lshl_b64")
def _(w, i):
    r = (w.rs64(i.args[1]) << (w.rs32(i.args[2]) & 63)) & M64
    w.ws64(i.args[0], r)
    w.scc = int(r != 0)


@op("s_lshr_b64")
def _(w, i):
    r = w.rs64(i.args[1]) >> (w.rs32(i.args[2]) & 63)
    w.ws64(i.args[0], r)
    w.scc = int(r != 0)


@op("s_cselect_b32")
def _(w, i):
    w.ws32(i.args[0], w.rs32(i.args[1]) if w.scc else w.rs32(i.args[2]))


@op("s_cselect_b64")
def _(w, i):
    w.ws64(i.args[0], w.rs64(i.args[1]) if w.scc else w.rs64(i.args[2]))


def _saveexec(name, fn):
    def h(w, i):
        old = w.exec
        w.exec = fn(w.rs64(i.args[1]), old) & M64
        w.ws64(i.args[0], old)
        w.scc = int(w.exec != 0)
    OPS[name] = h


_saveexec("s_and_saveexec_b64", lambda s, e: s & e)
_saveexec("s_or_saveexec_b64", lambda s, e: s | e)
_saveexec("s_xor_saveexec_b64", lambda s, e: s ^ e)
_saveexec("s_andn2_saveexec_b64", lambda s, e: s & ~e)
_saveexec("s_andn1_saveexec_b64", lambda s, e: ~s & e)


@op("s_addk_i32")
def _(w, i):
    r, o = _ovf_add(w.rs32(i.args[0]), _sx(int(i.args[1], 0), 16) & M32)
    w.ws32(i.args[0], r)
    w.scc = int(o)


@op("s_mulk_i32")
def _(w, i):
    w.ws32(i.args[0], (_sx(w.rs32(i.args[0]), 32) * _sx(int(i.args[1], 0), 16)) & M32)


_CMP = {"eq": lambda a, b: a == b, "lg": lambda a, b: a != b, "ne": lambda a, b: a != b, "gt": lambda a, b: a > b, "ge": lambda a, b: a >= b,
        "lt": lambda a, b: a < b, "le": lambda a, b: a <= b}


def _scmp(w, i):
    m = re.match(r"s_cmp(k?)_(\w+)_([iu])(32|64)", i.op)
    k, c, sg, bits = m.group(1), m.group(2), m.group(3), int(m.group(4))
    a = w.rs32(i.args[0]) if bits == 32 else w.rs64(i.args[0])
    if k:
        b = int(i.args[1], 0)
        b = _sx(b, 16) & M32 if sg == "i" else b & 0xFFFF
    else:
        b = w.rs32(i.args[1]) if bits == 32 else w.rs64(i.args[1])
    if sg == "i":
        a, b = _sx(a, bits), _sx(b, bits)
    w.scc = int(_CMP[c](a, b))


for _c in _CMP:
    for _t in ("i32", "u32", "u64"):
        OPS[f"s_cmp_{_c}_{_t}"] = _scmp
        OPS[f"s_cmpk_{_c}_{_t}"] = _scmp


@op("s_bitcmp0_b32")
def _(w, i):
    w.scc = int((w.rs32(i.args[0]) >> (w.rs32(i.args[1]) & 31)) & 1 == 0)


@op("s_bitcmp1_b32")
def _(w, i):
    w.scc = int((w.rs32(i.args[0]) >> (w.rs32(i.args[1]) & 31)) & 1 == 1)


def _branch(cond):
    def h(w, i):
        if cond(w):
            w.pc = i.target
    return h


OPS["s_branch"] = _branch(lambda w: True)
OPS["s_cbranch_scc0"] = _branch(lambda w: w.scc == 0)
OPS["s_cbranch_scc1"] = _branch(lambda w: w.scc == 1)
OPS["s_cbranch_vccz"] = _branch(lambda w: w.vcc == 0)
OPS["s_cbranch_vccnz"] = _branch(lambda w: w.vcc != 0)
OPS["s_cbranch_execz"] = _branch(lambda w: w.exec == 0)
OPS["s_cbranch_execnz"] = _branch(lambda w: w.exec != 0)


def _sload(n):
    def h(w, i):
        base = w.rs64(i.args[1])
        off = w.rs32(i.args[2])
        raw = bytes(_span(base + off, 4 * n))
        vals = struct.unpack(f"<{n}I", raw)
        if n == 1:
            w.ws32(i.args[0], vals[0])
        else:
            a = int(_RE_SR.match(i.args[0]).group(1))
            for k, v in enumerate(vals):
                w.s[a + k] = v
    return h


for _n, _name in ((1, "s_load_dword"), (2, "s_load_dwordx2"), (4, "s_load_dwordx4"), (8, "s_load_dwordx8"), (16, "s_load_dwordx16")):
    OPS[_name] = _sload(_n)


# ---- VALU -------------------------------------------------------------------------------------------------------------------------

def _dpp_source(w, i, src):
    """DPP on src0: (permuted value [64], write mask).  A lane whose source lane is outside its row / the wavefront or is not
    active (EXEC) gets 0 with bound_ctrl:1 and is not written otherwise; row_mask / bank_mask disable whole rows / banks."""
    mods = i.mods
    lane = LANES
    row, in_row = lane >> 4, lane & 15
    frm = np.full(64, -1, dtype=np.int64)
    if "quad_perm" in mods:
        q = [int(x) for x in mods["quad_perm"].strip("[]").split(",")]
        frm = (lane & ~3) | np.array(q, dtype=np.int64)[lane & 3]
    elif "row_shl" in mods:
        s = in_row + int(mods["row_shl"])
        frm = np.where(s < 16, row * 16 + s, -1)
    elif "row_shr" in mods:
        s = in_row - int(mods["row_shr"])
        frm = np.where(s >= 0, row * 16 + s, -1)
    elif "row_ror" in mods:
        frm = row * 16 + ((in_row - int(mods["row_ror"])) & 15)
    elif "row_bcast" in mods:
        if mods["row_bcast"] == "15":
            frm = np.where(row > 0, (row - 1) * 16 + 15, -1)
        elif mods["row_bcast"] == "31":
            frm = np.where(row >= 2, 31, -1)
        else:
            raise Unsupported(i.text)
    elif "row_mirror" in mods:
        frm = row * 16 + (15 - in_row)
    elif "row_half_mirror" in mods:
        frm = row * 16 + ((in_row & 8) | (7 - (in_row & 7)))
    elif "wave_shr" in mods:
        frm = np.where(lane >= 1, lane - 1, -1)
    elif "wave_shl" in mods:
        frm = np.where(lane < 63, lane + 1, -1)
    elif "wave_ror" in mods:
        frm = (lane - 1) & 63
    elif "wave_rol" in mods:
        frm = (lane + 1) & 63
    else:
        raise Unsupported("DPP control: " + i.text)
    execm = mask_of(w.exec)
    src = np.broadcast_to(np.asarray(src, dtype=np.uint32), (64,))
    valid = (frm >= 0) & execm[np.clip(frm, 0, 63)]
    val = np.where(valid, src[np.clip(frm, 0, 63)], np.uint32(0))
    rm, bm = int(mods.get("row_mask", "0xf"), 0), int(mods.get("bank_mask", "0xf"), 0)
    enabled = (((rm >> row) & 1) == 1) & (((bm >> (in_row >> 2)) & 1) == 1)
    bound = str(mods.get("bound_ctrl", "0")) in ("1", "0x1", "True") or mods.get("bound_ctrl") is True
    write = execm & enabled & (valid | bound)
    return val, write


_DPP_KEYS = ("quad_perm", "row_shl", "row_shr", "row_ror", "row_bcast", "row_mirror", "row_half_mirror", "wave_shr", "wave_shl", "wave_ror", "wave_rol")


_SDWA_SEL = {"BYTE_0": (0, 8), "BYTE_1": (8, 8), "BYTE_2": (16, 8), "BYTE_3": (24, 8), "WORD_0": (0, 16), "WORD_1": (16, 16), "DWORD": (0, 32)}


def _sdwa_src(x, sel, sext=False):
    sh, bits = _SDWA_SEL[sel]
    x = np.broadcast_to(np.asarray(x, dtype=np.uint32), (64,))
    if bits == 32:
        return x
    f = (x >> np.uint32(sh)) & np.uint32((1 << bits) - 1)
    if sext:  # sext(vN): the selected byte / word is sign-extended
        f = np.where(f >> np.uint32(bits - 1), f | np.uint32((0xFFFFFFFF << bits) & 0xFFFFFFFF), f)
    return f


def _valu(nsrc, fn, name=None):
    def h(w, i):
        srcs = [w.rv32(a[5:-1] if a.startswith("sext(") else a) for a in i.args[1:1 + nsrc]]
        mask = None
        if i.op.endswith("_dpp"):
            srcs[0], mask = _dpp_source(w, i, srcs[0])
        if i.op.endswith("_sdwa"):  # sub-dword source selection (zero-extended), full-dword destination only
            if any(a.startswith(("-", "|")) for a in i.args[1:]):
                raise Unsupported(i.text)
            srcs = [_sdwa_src(x, i.mods.get(f"src{k}_sel", "DWORD"), i.args[1 + k].startswith("sext(")) for k, x in enumerate(srcs)]
            if i.mods.get("dst_sel", "DWORD") != "DWORD":
                # the result's low byte / word goes to the selected place; the rest of the destination: zeros (UNUSED_PAD) or kept (UNUSED_PRESERVE)
                sh, bits = _SDWA_SEL[i.mods["dst_sel"]]
                field = np.uint32(((1 << bits) - 1) << sh)
                r = (np.broadcast_to(fn(*[np.asarray(s_, dtype=np.uint32) for s_ in srcs]), (64,)).astype(np.uint32) << np.uint32(sh)) & field
                how = i.mods.get("dst_unused", "UNUSED_PAD")
                if how == "UNUSED_PRESERVE":
                    r = r | (np.asarray(w.rv32(i.args[0]), dtype=np.uint32) & ~field)
                elif how != "UNUSED_PAD":
                    raise Unsupported(i.text)
                w.wv32(i.args[0], r, mask)
                return
        w.wv32(i.args[0], fn(*[np.asarray(s, dtype=np.uint32) for s in srcs]), mask)
    return h


def _reg(names, nsrc, fn):
    for n in names.split():
        h = _valu(nsrc, fn)
        for suffix in ("", "_e32", "_e64", "_dpp", "_sdwa"):
            OPS[n + suffix] = h


with np.errstate(over="ignore"):
    pass

_reg("v_mov_b32", 1, lambda a: a)
_reg("v_not_b32", 1, lambda a: ~a)
_reg("v_add_u32", 2, lambda a, b: a + b)
_reg("v_sub_u32", 2, lambda a, b: a - b)
_reg("v_subrev_u32", 2, lambda a, b: b - a)
_reg("v_and_b32", 2, lambda a, b: a & b)
_reg("v_or_b32", 2, lambda a, b: a | b)
_reg("v_xor_b32", 2, lambda a, b: a ^ b)
_reg("v_lshlrev_b32", 2, lambda a, b: b << (a & np.uint32(31)))
_reg("v_lshrrev_b32", 2, lambda a, b: b >> (a & np.uint32(31)))
_reg("v_ashrrev_i32", 2, lambda a, b: (b.view(np.int32) >> (a & np.uint32(31)).astype(np.int32)).view(np.uint32))
_reg("v_lshrrev_b16", 2, lambda a, b: (b & np.uint32(0xFFFF)) >> (a & np.uint32(15)))
_reg("v_mul_lo_u32", 2, lambda a, b: (a.astype(np.uint64) * b.astype(np.uint64)).astype(np.uint32))
_reg("v_mul_hi_u32", 2, lambda a, b: ((a.astype(np.uint64) * b.astype(np.uint64)) >> np.uint64(32)).astype(np.uint32))
_reg("v_mul_u32_u24", 2, lambda a, b: ((a & np.uint32(0xFFFFFF)).astype(np.uint64) * (b & np.uint32(0xFFFFFF)).astype(np.uint64)).astype(np.uint32))
_reg("v_min_u32", 2, lambda a, b: np.minimum(a, b))
_reg("v_max_u32", 2, lambda a, b: np.maximum(a, b))
_reg("v_bfi_b32", 3, lambda m, a, b: (m & a) | (~m & b))
_reg("v_alignbit_b32", 3, lambda hi, lo, s: (((hi.astype(np.uint64) << np.uint64(32)) | lo.astype(np.uint64)) >> (s & np.uint32(31)).astype(np.uint64)).astype(np.uint32))
_reg("v_lshl_add_u32", 3, lambda a, s, c: (a << (s & np.uint32(31))) + c)
_reg("v_add_lshl_u32", 3, lambda a, b, s: (a + b) << (s & np.uint32(31)))
_reg("v_lshl_or_b32", 3, lambda a, s, c: (a << (s & np.uint32(31))) | c)
_reg("v_and_or_b32", 3, lambda a, b, c: (a & b) | c)
_reg("v_add3_u32", 3, lambda a, b, c: a + b + c)
_reg("v_or3_b32", 3, lambda a, b, c: a | b | c)
_reg("v_xad_u32", 3, lambda a, b, c: (a ^ b) + c)
_reg("v_mad_u32_u24", 3, lambda a, b, c: ((a & np.uint32(0xFFFFFF)).astype(np.uint64) * (b & np.uint32(0xFFFFFF)).astype(np.uint64)).astype(np.uint32) + c)
_reg("v_bfe_u32", 3, lambda a, o, n: np.where((n & np.uint32(31)) == 0, np.uint32(0), (a >> (o & np.uint32(31))) & ((np.uint64(1) << (n & np.uint32(31)).astype(np.uint64)) - np.uint64(1)).astype(np.uint32)))
_reg("v_bcnt_u32_b32", 2, lambda a, b: np.array([bin(int(x)).count("1") for x in np.broadcast_to(a, (64,))], dtype=np.uint32) + b)


def _bfe_i32(a, o, n):
    o = (o & np.uint32(31)).astype(np.int64)
    n = (n & np.uint32(31)).astype(np.int64)
    # (S0 is SIGNED: the shift is arithmetic, so a field that runs past bit 31 is filled with copies of the sign -- LLVM folds
    # sbfe(x, off, n) with off + n >= 32 to ashr(x, off); tools/fuzz_interpreter_vs_compiler.py found the logical shift here)
    x = (np.broadcast_to(a, (64,)).astype(np.uint32).view(np.int32).astype(np.int64) >> o) & ((np.int64(1) << n) - 1)
    sign = (x >> np.maximum(n - 1, 0)) & 1
    x = np.where((n > 0) & (sign == 1), x - (np.int64(1) << n), x)
    return np.where(n == 0, 0, x).astype(np.int64).astype(np.uint32)


_reg("v_bfe_i32", 3, _bfe_i32)


def _perm(hi, lo, sel):
    """v_perm_b32 D, S0, S1, S2: byte k of D = byte sel[k] of {S0, S1} (0-3: S1 = `lo`, 4-7: S0 = `hi`; 8-11 sign bytes, 12 0x00, >= 13 0xff)"""
    hi, lo, sel = (np.broadcast_to(x, (64,)).astype(np.uint64) for x in (hi, lo, sel))
    both = (hi << np.uint64(32)) | lo
    sb = np.ascontiguousarray(sel.astype(np.uint32)).view(np.uint8).reshape(64, 4)
    if int(sb.max()) <= 7:  # (the only selectors the kernels use: plain byte picks)
        bb = np.ascontiguousarray(both).view(np.uint8).reshape(64, 8)
        return np.ascontiguousarray(np.take_along_axis(bb, sb.astype(np.int64), axis=1)).view(np.uint32).reshape(64)
    out = np.zeros(64, dtype=np.uint64)
    for k in range(4):
        s = (sel >> np.uint64(8 * k)) & np.uint64(0xFF)
        byte = (both >> ((s & np.uint64(7)) * np.uint64(8))) & np.uint64(0xFF)
        sign_src = np.select([s == 8, s == 9, s == 10, s == 11], [lo >> np.uint64(15), lo >> np.uint64(31), hi >> np.uint64(15), hi >> np.uint64(31)], 0) & np.uint64(1)
        byte = np.where(s <= 7, byte, np.where(s <= 11, sign_src * np.uint64(0xFF), np.where(s == 12, np.uint64(0), np.uint64(0xFF))))
        out |= byte << np.uint64(8 * k)
    return out.astype(np.uint32)


_reg("v_perm_b32", 3, _perm)


@op("v_bitop3_b32")
def _(w, i):
    t = int(i.mods["bitop3"], 0)
    a, b, c = (np.broadcast_to(np.asarray(w.rv32(x), dtype=np.uint32), (64,)) for x in i.args[1:4])
    out = np.zeros(64, dtype=np.uint32)
    for k in range(8):  # truth-table bit k: the result where (a, b, c) = ((k >> 2) & 1, (k >> 1) & 1, k & 1)
        if (t >> k) & 1:
            out |= (a if k & 4 else ~a) & (b if k & 2 else ~b) & (c if k & 1 else ~c)
    w.wv32(i.args[0], out)


def _cndmask(w, i):
    sel = mask_of(w.rs64(i.args[3]) if len(i.args) > 3 else w.vcc)

    def src(tok):  # (the VOP3 float modifiers act on the bit pattern: -x flips bit 31, |x| clears it; LLVM selects x ^ 0x80000000 this way)
        neg = tok.startswith("-") and not tok[1:2].isdigit()   # (-1 is an inline constant, -v4 a modifier)
        tok = tok[1:] if neg else tok
        ab = tok.startswith("|") and tok.endswith("|")
        tok = tok[1:-1] if ab else tok
        x = np.asarray(w.rv32(tok), dtype=np.uint32)
        x = x & np.uint32(0x7FFFFFFF) if ab else x
        return x ^ np.uint32(0x80000000) if neg else x

    mask = None
    if i.op.endswith("_sdwa"):  # sub-dword sources (as _valu: zero- or sign-extended selection), the result's low field to dst_sel
        if any(t.startswith(("-", "|")) for t in i.args[1:3]):
            raise Unsupported(i.text)
        a, b = (_sdwa_src(w.rv32(t[5:-1] if t.startswith("sext(") else t), i.mods.get(f"src{k}_sel", "DWORD"), t.startswith("sext("))
                for k, t in enumerate(i.args[1:3]))
        r = np.where(sel, b, a).astype(np.uint32)
        if i.mods.get("dst_sel", "DWORD") != "DWORD":
            sh, bits = _SDWA_SEL[i.mods["dst_sel"]]
            field = np.uint32(((1 << bits) - 1) << sh)
            r = (r << np.uint32(sh)) & field
            how = i.mods.get("dst_unused", "UNUSED_PAD")
            if how == "UNUSED_PRESERVE":
                r = r | (np.asarray(w.rv32(i.args[0]), dtype=np.uint32) & ~field)
            elif how != "UNUSED_PAD":
                raise Unsupported(i.text)
        w.wv32(i.args[0], r)
        return
    a, b = src(i.args[1]), src(i.args[2])
    if i.op.endswith("_dpp"):
        a, mask = _dpp_source(w, i, a)
    w.wv32(i.args[0], np.where(sel, b, a), mask)


for _s in ("_e32", "_e64", "_dpp", "_sdwa", ""):
    OPS["v_cndmask_b32" + _s] = _cndmask


def _carry_op(sub, rev, with_carry_in):
    def h(w, i):
        # vD, carry-out (vcc | s[a:b]), src0, src1 [, carry-in]
        a = np.broadcast_to(np.asarray(w.rv32(i.args[2]), dtype=np.uint32), (64,)).astype(np.uint64)
        b = np.broadcast_to(np.asarray(w.rv32(i.args[3]), dtype=np.uint32), (64,)).astype(np.uint64)
        mask = None
        if i.op.endswith("_dpp"):  # (DPP permutes src0)
            a32, mask = _dpp_source(w, i, a.astype(np.uint32))
            a = a32.astype(np.uint64)
        if rev:
            a, b = b, a
        cin = mask_of(w.rs64(i.args[4])).astype(np.uint64) if with_carry_in else np.uint64(0)
        if sub:
            r = a - b - cin
            carry = (b + cin) > a
        else:
            r = a + b + cin
            carry = r > np.uint64(M32)
        w.wv32(i.args[0], (r & np.uint64(M32)).astype(np.uint32), mask)
        w.wmask(i.args[1], carry)
    return h


for _s in ("_e32", "_e64", "_dpp"):
    OPS["v_add_co_u32" + _s] = _carry_op(False, False, False)
    OPS["v_sub_co_u32" + _s] = _carry_op(True, False, False)
    OPS["v_subrev_co_u32" + _s] = _carry_op(True, True, False)
    OPS["v_addc_co_u32" + _s] = _carry_op(False, False, True)
    OPS["v_subb_co_u32" + _s] = _carry_op(True, False, True)
    OPS["v_subbrev_co_u32" + _s] = _carry_op(True, True, True)


@op("v_lshl_add_u64")
def _(w, i):
    a, s, c = w.rv64(i.args[1]), np.uint64(w.rs32(i.args[2]) & 7), w.rv64(i.args[3])
    w.wv64(i.args[0], (np.asarray(a, dtype=np.uint64) << s) + np.asarray(c, dtype=np.uint64))


@op("v_lshlrev_b64")
def _(w, i):
    sh = np.broadcast_to(np.asarray(w.rv32(i.args[1]), dtype=np.uint32), (64,)).astype(np.uint64) & np.uint64(63)
    w.wv64(i.args[0], np.broadcast_to(np.asarray(w.rv64(i.args[2]), dtype=np.uint64), (64,)) << sh)


@op("v_lshrrev_b64")
def _(w, i):
    sh = np.broadcast_to(np.asarray(w.rv32(i.args[1]), dtype=np.uint32), (64,)).astype(np.uint64) & np.uint64(63)
    w.wv64(i.args[0], np.broadcast_to(np.asarray(w.rv64(i.args[2]), dtype=np.uint64), (64,)) >> sh)


@op("v_mad_u64_u32")
def _(w, i):
    a = np.broadcast_to(np.asarray(w.rv32(i.args[2]), dtype=np.uint32), (64,)).astype(np.uint64)
    b = np.broadcast_to(np.asarray(w.rv32(i.args[3]), dtype=np.uint32), (64,)).astype(np.uint64)
    c = np.broadcast_to(np.asarray(w.rv64(i.args[4]), dtype=np.uint64), (64,))
    p = a * b
    r = p + c
    w.wv64(i.args[0], r)
    w.wmask(i.args[1], r < p)


# ---- instructions the product's kernels do not contain; here so that tools/fuzz_interpreter_vs_compiler.py can run what the
# compiler emits for arbitrary integer HIP code (each one is held against the compiler's own output by that tool) ----

def _ffbh_u32(a):
    a = np.broadcast_to(a, (64,)).astype(np.uint64)
    out = np.full(64, 0xFFFFFFFF, dtype=np.uint32)
    for k in range(64):
        v = int(a[k])
        if v:
            out[k] = 32 - v.bit_length()
    return out


def _ffbl_b32(a):
    a = np.broadcast_to(a, (64,)).astype(np.uint64)
    out = np.full(64, 0xFFFFFFFF, dtype=np.uint32)
    for k in range(64):
        v = int(a[k])
        if v:
            out[k] = (v & -v).bit_length() - 1
    return out


def _bfrev(a):
    a = np.broadcast_to(a, (64,)).astype(np.uint32).copy()
    a = ((a >> np.uint32(1)) & np.uint32(0x55555555)) | ((a & np.uint32(0x55555555)) << np.uint32(1))
    a = ((a >> np.uint32(2)) & np.uint32(0x33333333)) | ((a & np.uint32(0x33333333)) << np.uint32(2))
    a = ((a >> np.uint32(4)) & np.uint32(0x0F0F0F0F)) | ((a & np.uint32(0x0F0F0F0F)) << np.uint32(4))
    return a.byteswap()


def _s24(x):
    x = np.broadcast_to(x, (64,)).astype(np.int64) & 0xFFFFFF
    return np.where(x & 0x800000, x - 0x1000000, x)


def _si32(x):
    return np.broadcast_to(x, (64,)).astype(np.uint32).view(np.int32).astype(np.int64)


_reg("v_ffbh_u32", 1, _ffbh_u32)
_reg("v_ffbl_b32", 1, _ffbl_b32)
_reg("v_bfrev_b32", 1, _bfrev)
_reg("v_xnor_b32", 2, lambda a, b: ~(a ^ b))
_reg("v_alignbyte_b32", 3, lambda hi, lo, s: (((hi.astype(np.uint64) << np.uint64(32)) | lo.astype(np.uint64)) >> (np.uint64(8) * (s & np.uint32(3)).astype(np.uint64))).astype(np.uint32))
_reg("v_max_i32", 2, lambda a, b: np.maximum(_si32(a), _si32(b)).astype(np.uint32))
_reg("v_min_i32", 2, lambda a, b: np.minimum(_si32(a), _si32(b)).astype(np.uint32))
_reg("v_mul_hi_i32", 2, lambda a, b: ((_si32(a) * _si32(b)) >> 32).astype(np.uint32))
_reg("v_mul_i32_i24", 2, lambda a, b: (_s24(a) * _s24(b)).astype(np.uint32))
_reg("v_mul_hi_i32_i24", 2, lambda a, b: ((_s24(a) * _s24(b)) >> 32).astype(np.uint32))
_reg("v_mul_hi_u32_u24", 2, lambda a, b: (((a & np.uint32(0xFFFFFF)).astype(np.uint64) * (b & np.uint32(0xFFFFFF)).astype(np.uint64)) >> np.uint64(32)).astype(np.uint32))
_reg("v_mad_i32_i24", 3, lambda a, b, c: ((_s24(a) * _s24(b)).astype(np.uint32) + c))


# 16-bit VALU (gfx9: the result's high 16 bits are written as zero -- the compiler drops zero-extensions on that account; the "legacy"
# multiply-adds are the forms that do the same in VOP3)
_H = np.uint32(0xFFFF)


def _s16(x):
    x = (np.broadcast_to(x, (64,)).astype(np.int64)) & 0xFFFF
    return np.where(x & 0x8000, x - 0x10000, x)


_reg("v_add_u16", 2, lambda a, b: (a + b) & _H)
_reg("v_sub_u16", 2, lambda a, b: (a - b) & _H)
_reg("v_subrev_u16", 2, lambda a, b: (b - a) & _H)
_reg("v_mul_lo_u16", 2, lambda a, b: ((a & _H) * (b & _H)) & _H)
_reg("v_max_u16", 2, lambda a, b: np.maximum(a & _H, b & _H))
_reg("v_min_u16", 2, lambda a, b: np.minimum(a & _H, b & _H))
_reg("v_max_i16", 2, lambda a, b: np.maximum(_s16(a), _s16(b)).astype(np.uint32) & _H)
_reg("v_min_i16", 2, lambda a, b: np.minimum(_s16(a), _s16(b)).astype(np.uint32) & _H)
_reg("v_lshlrev_b16", 2, lambda a, b: ((b & _H) << (a & np.uint32(15))) & _H)
_reg("v_ashrrev_i16", 2, lambda a, b: (_s16(b) >> (np.broadcast_to(a, (64,)).astype(np.int64) & 15)).astype(np.uint32) & _H)
_reg("v_mad_legacy_u16", 3, lambda a, b, c: ((a & _H) * (b & _H) + (c & _H)) & _H)
_reg("v_mad_legacy_i16", 3, lambda a, b, c: (_s16(a) * _s16(b) + _s16(c)).astype(np.uint32) & _H)


def _vcmp16(w, i):
    m = re.match(r"v_cmp_(\w+?)_([iu])16", i.op)
    c, sg = m.group(1), m.group(2)
    a = np.broadcast_to(np.asarray(w.rv32(i.args[1]), dtype=np.uint32), (64,))
    b = np.broadcast_to(np.asarray(w.rv32(i.args[2]), dtype=np.uint32), (64,))
    a, b = (_s16(a), _s16(b)) if sg == "i" else ((a & _H).astype(np.int64), (b & _H).astype(np.int64))
    w.wmask(i.args[0], _CMP[c](a, b))


for _c in ("lt", "le", "gt", "ge", "eq", "ne"):
    for _t in ("i16", "u16"):
        for _s in ("_e32", "_e64"):
            OPS[f"v_cmp_{_c}_{_t}{_s}"] = _vcmp16


@op("v_ashrrev_i64")
def _(w, i):
    sh = np.broadcast_to(np.asarray(w.rv32(i.args[1]), dtype=np.uint32), (64,)).astype(np.int64) & np.int64(63)
    w.wv64(i.args[0], (np.broadcast_to(np.asarray(w.rv64(i.args[2]), dtype=np.uint64), (64,)).view(np.int64) >> sh).view(np.uint64))


def _s_bfe_i32(a, b, c):
    off, n = b & 31, (b >> 16) & 127
    if n == 0:
        return 0, False
    f = (_sx(a, 32) >> off) & ((1 << n) - 1)      # (signed source: the shift is arithmetic, as in V_BFE_I32)
    r = _sx(f, n) & M32 if n < 32 else f & M32
    return r, r != 0


_salu2("s_bfe_i32", 32, _s_bfe_i32)


@op("s_bfe_i64", "s_bfe_u64")
def _(w, i):
    # S0 is 64 bits wide, S1 one dword: offset = S1[5:0], width = S1[22:16]; a field that runs past bit 63 ends there
    a, b = w.rs64(i.args[1]), w.rs32(i.args[2])
    off, n = b & 63, min((b >> 16) & 127, 64)
    if n == 0:
        r = 0
    elif i.op == "s_bfe_i64":
        f = (_sx(a, 64) >> off) & ((1 << n) - 1)      # (arithmetic shift: sign bits fill a field that reaches past bit 63)
        r = _sx(f, n) & M64
    else:
        r = (a >> off) & ((1 << n) - 1)
    w.ws64(i.args[0], r)
    w.scc = int(r != 0)


@op("s_bitset1_b32", "s_bitset0_b32")
def _(w, i):
    bit = 1 << (w.rs32(i.args[1]) & 31)
    old = w.rs32(i.args[0])
    w.ws32(i.args[0], (old | bit) if i.op == "s_bitset1_b32" else (old & ~bit & M32))


@op("s_ashr_i64")
def _(w, i):
    r = (_sx(w.rs64(i.args[1]), 64) >> (w.rs32(i.args[2]) & 63)) & M64
    w.ws64(i.args[0], r)
    w.scc = int(r != 0)


@op("s_bcnt1_i32_b32")
def _(w, i):
    r = bin(w.rs32(i.args[1])).count("1")
    w.ws32(i.args[0], r)
    w.scc = int(r != 0)


@op("s_sext_i32_i16")
def _(w, i):
    w.ws32(i.args[0], _sx(w.rs32(i.args[1]) & 0xFFFF, 16) & M32)


@op("s_sext_i32_i8")
def _(w, i):
    w.ws32(i.args[0], _sx(w.rs32(i.args[1]) & 0xFF, 8) & M32)


@op("v_mad_i64_i32")
def _(w, i):
    a, b = _si32(np.asarray(w.rv32(i.args[2]), dtype=np.uint32)), _si32(np.asarray(w.rv32(i.args[3]), dtype=np.uint32))
    c = np.broadcast_to(np.asarray(w.rv64(i.args[4]), dtype=np.uint64), (64,))
    p = (a * b).view(np.uint64)
    r = p + c
    w.wv64(i.args[0], r)
    # (the carry-out pair is written by the instruction; the compiler never reads it after a signed multiply-add)
    w.wmask(i.args[1], np.zeros(64, dtype=bool))


@op("s_flbit_i32_b64")
def _(w, i):
    v = w.rs64(i.args[1])
    w.ws32(i.args[0], 64 - v.bit_length() if v else M32)


@op("s_flbit_i32_b32")
def _(w, i):
    v = w.rs32(i.args[1])
    w.ws32(i.args[0], 32 - v.bit_length() if v else M32)


@op("v_mov_b64_e32", "v_mov_b64")
def _(w, i):
    w.wv64(i.args[0], np.broadcast_to(np.asarray(w.rv64(i.args[1]), dtype=np.uint64), (64,)))


@op("v_pk_mov_b32")
def _(w, i):
    # D[31:0] = OPSEL[0] ? S0[63:32] : S0[31:0];  D[63:32] = OPSEL[1] ? S1[63:32] : S1[31:0]   (CDNA3 ISA; op_sel_hi plays no part:
    # LLVM's own 64-bit move through this instruction is op_sel:[0,1], and every use in these kernels -- the copy-out's straddle of
    # two LDS vectors, op_sel:[1,0] = {S0.hi, S1.lo} -- reads that way)
    sel = [int(x) for x in i.mods.get("op_sel", "[0,0]").strip("[]").split(",")]
    s0 = np.broadcast_to(np.asarray(w.rv64(i.args[1]), dtype=np.uint64), (64,))
    s1 = np.broadcast_to(np.asarray(w.rv64(i.args[2]), dtype=np.uint64), (64,))
    lo = (s0 >> np.uint64(32 * sel[0])) & np.uint64(M32)
    hi = (s1 >> np.uint64(32 * sel[1])) & np.uint64(M32)
    w.wv64(i.args[0], lo | (hi << np.uint64(32)))


def _vcmp(w, i):
    m = re.match(r"v_cmp(x?)_(\w+?)_([iu])(32|64)", i.op)
    x, c, sg, bits = m.group(1), m.group(2), m.group(3), int(m.group(4))
    rd = w.rv32 if bits == 32 else w.rv64
    dt = np.uint32 if bits == 32 else np.uint64
    a = np.broadcast_to(np.asarray(rd(i.args[1]), dtype=dt), (64,))
    b = np.broadcast_to(np.asarray(rd(i.args[2]), dtype=dt), (64,))
    if sg == "i":
        a, b = a.view(np.int32 if bits == 32 else np.int64), b.view(np.int32 if bits == 32 else np.int64)
    r = _CMP[c](a, b)
    w.wmask(i.args[0], r)
    if x:
        w.exec = bits_of(r & mask_of(w.exec))


for _c in _CMP:
    for _t in ("i32", "u32", "i64", "u64"):
        for _s in ("_e32", "_e64"):
            OPS[f"v_cmp_{_c}_{_t}{_s}"] = _vcmp
            OPS[f"v_cmpx_{_c}_{_t}{_s}"] = _vcmp


@op("v_readfirstlane_b32")
def _(w, i):
    src = np.broadcast_to(np.asarray(w.rv32(i.args[1]), dtype=np.uint32), (64,))
    lane = (w.exec & -w.exec).bit_length() - 1 if w.exec else 0
    w.ws32(i.args[0], int(src[lane]))


@op("v_readlane_b32")
def _(w, i):
    w.ws32(i.args[0], int(w.rv32(i.args[1])[w.rs32(i.args[2]) & 63]))


@op("v_writelane_b32")
def _(w, i):
    n = int(_RE_V.match(i.args[0]).group(1))
    w.v[n, w.rs32(i.args[2]) & 63] = w.rs32(i.args[1])


@op("v_mbcnt_lo_u32_b32")
def _(w, i):
    m = np.broadcast_to(np.asarray(w.rv32(i.args[1]), dtype=np.uint32), (64,))  # (a mask per lane when the source is a VGPR)
    cnt = np.array([bin(int(m[l]) & ((1 << min(l, 32)) - 1)).count("1") for l in range(64)], dtype=np.uint32)
    w.wv32(i.args[0], cnt + np.asarray(w.rv32(i.args[2]), dtype=np.uint32))


@op("v_mbcnt_hi_u32_b32")
def _(w, i):
    m = np.broadcast_to(np.asarray(w.rv32(i.args[1]), dtype=np.uint32), (64,))
    cnt = np.array([bin(int(m[l]) & ((1 << max(l - 32, 0)) - 1)).count("1") for l in range(64)], dtype=np.uint32)
    w.wv32(i.args[0], cnt + np.asarray(w.rv32(i.args[2]), dtype=np.uint32))


# the float instructions of LLVM's unsigned-division expansion (reciprocal estimate + integer correction steps: any reciprocal
# within an ulp gives the same quotient)
def _f(x):
    return np.broadcast_to(np.asarray(x, dtype=np.uint32), (64,)).view(np.float32)


def _fu(x):
    return np.asarray(x, dtype=np.float32).view(np.uint32)


def _quiet(fn):
    def g(*a):
        with np.errstate(all="ignore"):
            return fn(*a)
    return g


_reg("v_cvt_f32_u32", 1, lambda a: _fu(np.broadcast_to(a, (64,)).astype(np.float32)))
for _k in range(4):  # (the divisor of LLVM's division expansion when it is known to fit a byte)
    _reg(f"v_cvt_f32_ubyte{_k}", 1, (lambda k: lambda a: _fu(((np.broadcast_to(a, (64,)) >> np.uint32(8 * k)) & np.uint32(0xFF)).astype(np.float32)))(_k))
_reg("v_rcp_iflag_f32 v_rcp_f32", 1, _quiet(lambda a: _fu(np.float32(1.0) / _f(a))))
_reg("v_mul_f32", 2, _quiet(lambda a, b: _fu(_f(a) * _f(b))))
_reg("v_trunc_f32", 1, lambda a: _fu(np.trunc(_f(a))))


def _cvt_u32_f32(a):
    f = _f(a).astype(np.float64)
    f = np.where(np.isnan(f), 0.0, np.clip(np.trunc(f), 0.0, 4294967295.0))
    return f.astype(np.uint64).astype(np.uint32)


_reg("v_cvt_u32_f32", 1, _quiet(_cvt_u32_f32))


@op("v_fmac_f32_e32", "v_fmac_f32_e64", "v_fmac_f32")
def _(w, i):
    n = int(_RE_V.match(i.args[0]).group(1))
    with np.errstate(all="ignore"):  # (lanes outside EXEC hold junk)
        r = (_f(w.rv32(i.args[1])).astype(np.float64) * _f(w.rv32(i.args[2])).astype(np.float64) + _f(w.v[n]).astype(np.float64)).astype(np.float32)
    w.wv32(i.args[0], _fu(r))


# ---- LDS ----------------------------------------------------------------------------------------------------------------------------

# Bank-conflict pricing of the LDS instructions that were executed (tools/dynamic_profile.py), after the MI355X guide's LDS table: a
# wave64 access is served in fixed lane groups, one LDS cycle per group when conflict-free; only lanes of one group conflict,
# identical dword addresses broadcast, every further distinct address on a busy bank adds a cycle.  (kernel, opcode) ->
# [instructions, LDS-array cycles, conflict-free cycles, pipe-busy cycles = max(array, issue) per instruction]
LDS_PROFILE = None
_G2x32 = [np.arange(0, 32), np.arange(32, 64)]
_G4x16 = [np.arange(16 * g, 16 * g + 16) for g in range(4)]
_G8x8 = [np.arange(8 * g, 8 * g + 8) for g in range(8)]
_GR128 = [np.array(x) for x in ([0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31],
                                [32, 33, 34, 35, 44, 45, 46, 47, 52, 53, 54, 55, 56, 57, 58, 59], [36, 37, 38, 39, 40, 41, 42, 43, 48, 49, 50, 51, 60, 61, 62, 63])]
_LDS_RULES = {  # opcode: (lane groups, banks, issue cycles)
    "ds_read_b32": (_G2x32, 32, 2), "ds_read_b64": (_G2x32, 64, 2), "ds_read_b128": (_GR128, 64, 4), "ds_read_b96": (_G8x8, 32, 8),
    "ds_write_b32": (_G2x32, 32, 4), "ds_write_b64": (_G4x16, 32, 6), "ds_write_b128": (_G8x8, 32, 13), "ds_write_b96": (_G8x8, 32, 10),
    "ds_read2_b32": (_G2x32, 32, 2), "ds_read2st64_b32": (_G2x32, 32, 2), "ds_read2_b64": (_G4x16, 32, 4), "ds_read2st64_b64": (_G4x16, 32, 4),
    "ds_write2_b32": (_G2x32, 32, 4), "ds_write2st64_b32": (_G2x32, 32, 4), "ds_write2_b64": (_G4x16, 32, 6), "ds_write2st64_b64": (_G4x16, 32, 6)}


def _lds_price(w, op, addr, nbytes, act):
    rule = _LDS_RULES.get(op)
    if LDS_PROFILE is None or rule is None:
        return
    groups, banks, issue = rule
    dwords = (addr.astype(np.int64)[:, None] + np.arange(0, nbytes, 4, dtype=np.int64)[None, :]) >> 2  # [lane, dwords per lane]
    cycles = 0
    for g in groups:
        lanes = g[act[g]]
        if lanes.size == 0:
            cycles += 1
            continue
        d = np.unique(dwords[lanes].reshape(-1))
        cycles += int(np.bincount(d % banks, minlength=banks).max())
    rec = LDS_PROFILE[(w.kernel_name, op)]
    rec[0] += 1
    rec[1] += cycles
    rec[2] += len(groups)
    rec[3] += max(cycles, issue)

def _lds_idx(w, addr, nbytes, what):
    lds = w.wg.lds
    a = np.broadcast_to(np.asarray(addr, dtype=np.uint32), (64,)).astype(np.int64)
    act = mask_of(w.exec)
    if act.any():
        hi = int(a[act].max()) + nbytes
        if hi > lds.size:
            raise Unsupported(f"{what}: LDS byte address {hi - nbytes} + {nbytes} beyond the {lds.size} bytes this launch allocated")
        if (a[act] % min(nbytes, 8 if nbytes == 8 else nbytes if nbytes < 16 else 16)).any() and nbytes >= 4:
            # (ds_read/write_b64 want 8-byte, b128 16-byte aligned addresses unless the unaligned mode is on; a misaligned b128 is a
            # 16x slower access on this part: the kernels never issue one on purpose)
            raise Unsupported(f"{what}: misaligned LDS address {int(a[act][(a[act] % 4 != 0).argmax()])}")
    a = np.where(act, a, 0)
    if LDS_PROFILE is not None:
        _lds_price(w, what.split(" ", 1)[0], a, nbytes, act)
    return a[:, None] + np.arange(nbytes, dtype=np.int64)[None, :], act


def _ds_read(nbytes):
    def h(w, i):
        off = int(i.mods.get("offset", "0"), 0)
        idx, act = _lds_idx(w, np.asarray(w.rv32(i.args[1]), dtype=np.uint32) + np.uint32(off), nbytes, i.text)
        data = np.ascontiguousarray(w.wg.lds[idx]).view(np.uint32)  # [64, nbytes / 4]
        n0, cnt = w.vrange(i.args[0])
        for k in range(cnt):
            w.v[n0 + k] = np.where(act, data[:, k], w.v[n0 + k])
    return h


def _ds_write(nbytes):
    def h(w, i):
        off = int(i.mods.get("offset", "0"), 0)
        idx, act = _lds_idx(w, np.asarray(w.rv32(i.args[0]), dtype=np.uint32) + np.uint32(off), nbytes, i.text)
        n0, cnt = w.vrange(i.args[1])
        data = np.ascontiguousarray(w.v[n0:n0 + cnt].T).view(np.uint8)  # [64, nbytes]
        w.wg.lds[idx[act]] = data[act]
    return h


for _b, _n in ((4, "b32"), (8, "b64"), (12, "b96"), (16, "b128")):
    OPS["ds_read_" + _n] = _ds_read(_b)
    OPS["ds_write_" + _n] = _ds_write(_b)


def _ds_read2(stride):
    def h(w, i):
        base = np.array(w.rv32(i.args[1]), dtype=np.uint32)  # (a copy: the destination range may contain the address register)
        n0, cnt = w.vrange(i.args[0])
        half = cnt // 2
        for part, key in enumerate(("offset0", "offset1")):
            off = int(i.mods.get(key, "0"), 0) * stride
            idx, act = _lds_idx(w, base + np.uint32(off), 4 * half, i.text)
            data = np.ascontiguousarray(w.wg.lds[idx]).view(np.uint32)
            for k in range(half):
                w.v[n0 + part * half + k] = np.where(act, data[:, k], w.v[n0 + part * half + k])
    return h


OPS["ds_read2_b32"] = _ds_read2(4)
OPS["ds_read2_b64"] = _ds_read2(8)
OPS["ds_read2st64_b32"] = _ds_read2(4 * 64)
OPS["ds_read2st64_b64"] = _ds_read2(8 * 64)


def _ds_write2(width, stride):
    def h(w, i):
        base = np.array(w.rv32(i.args[0]), dtype=np.uint32)
        for part, key in enumerate(("offset0", "offset1")):
            off = int(i.mods.get(key, "0"), 0) * stride
            idx, act = _lds_idx(w, base + np.uint32(off), width, i.text)
            n0, cnt = w.vrange(i.args[1 + part])
            data = np.ascontiguousarray(w.v[n0:n0 + width // 4].T).view(np.uint8)
            w.wg.lds[idx[act]] = data[act]
    return h


OPS["ds_write2_b32"] = _ds_write2(4, 4)
OPS["ds_write2_b64"] = _ds_write2(8, 8)
OPS["ds_write2st64_b32"] = _ds_write2(4, 4 * 64)
OPS["ds_write2st64_b64"] = _ds_write2(8, 8 * 64)


@op("ds_bpermute_b32")
def _(w, i):
    off = int(i.mods.get("offset", "0"), 0)
    addr = np.asarray(w.rv32(i.args[1]), dtype=np.uint32) + np.uint32(off)
    src_lane = ((np.broadcast_to(addr, (64,)) >> np.uint32(2)) & np.uint32(63)).astype(np.int64)
    data = np.broadcast_to(np.asarray(w.rv32(i.args[2]), dtype=np.uint32), (64,))
    act = mask_of(w.exec)
    w.wv32(i.args[0], np.where(act[src_lane], data[src_lane], np.uint32(0)))  # (an inactive source lane contributes 0)


@op("ds_permute_b32")
def _(w, i):
    # the push form: every ACTIVE lane sends its data to lane (addr >> 2) & 63; a lane nobody sends to reads 0; of several senders
    # to one lane the highest-numbered wins (not among the product's instructions: here for tools/fuzz_interpreter_vs_compiler.py)
    off = int(i.mods.get("offset", "0"), 0)
    addr = np.broadcast_to(np.asarray(w.rv32(i.args[1]), dtype=np.uint32), (64,)) + np.uint32(off)
    dst_lane = ((addr >> np.uint32(2)) & np.uint32(63)).astype(np.int64)
    data = np.broadcast_to(np.asarray(w.rv32(i.args[2]), dtype=np.uint32), (64,))
    act = mask_of(w.exec)
    out = np.zeros(64, dtype=np.uint32)
    for l in range(64):
        if act[l]:
            out[dst_lane[l]] = data[l]
    w.wv32(i.args[0], out)


# ---- global memory ------------------------------------------------------------------------------------------------------------------

def _gaddr(w, i, vtok, stok):
    off = _sx(int(i.mods.get("offset", "0"), 0), 13) if "offset" in i.mods else 0
    if stok == "off":
        a = np.broadcast_to(np.asarray(w.rv64(vtok), dtype=np.uint64), (64,))
    else:
        a = np.uint64(w.rs64(stok)) + np.broadcast_to(np.asarray(w.rv32(vtok), dtype=np.uint32), (64,)).astype(np.uint64)
    return (a.astype(np.int64) + off).astype(np.uint64)


def _gload(nd):
    def h(w, i):
        addr = _gaddr(w, i, i.args[1], i.args[2])
        act = mask_of(w.exec)
        if act.any() and (addr[act] % np.uint64(4)).any():
            raise Unsupported("misaligned global load: " + i.text)
        data = mem_gather(addr, act, 4 * nd).view(np.uint32)
        n0, cnt = w.vrange(i.args[0])
        for k in range(nd):
            w.v[n0 + k] = np.where(act, data[:, k], w.v[n0 + k])
    return h


def _gstore(nd):
    def h(w, i):
        addr = _gaddr(w, i, i.args[0], i.args[2])
        act = mask_of(w.exec)
        if act.any() and (addr[act] % np.uint64(4)).any():
            raise Unsupported("misaligned global store: " + i.text)
        n0, cnt = w.vrange(i.args[1])
        mem_scatter(addr, act, np.ascontiguousarray(w.v[n0:n0 + nd].T).view(np.uint8))
    return h


for _nd, _n in ((1, "dword"), (2, "dwordx2"), (3, "dwordx3"), (4, "dwordx4")):
    OPS["global_load_" + _n] = _gload(_nd)
    OPS["global_store_" + _n] = _gstore(_nd)


def _gatomic(fn, nd=1):
    def h(w, i):
        ret = len(i.args) == 4  # vdst, vaddr, vdata, saddr | vaddr, vdata, saddr
        a = i.args[1:] if ret else i.args
        addr = _gaddr(w, i, a[0], a[2])
        n0, _ = w.vrange(a[1])
        act = mask_of(w.exec)
        old = np.zeros((64, nd), dtype=np.uint32)
        for lane in np.nonzero(act)[0]:  # (lane order = the order the atomics of one instruction are performed in)
            cell = np.frombuffer((C.c_uint32 * nd).from_address(int(addr[lane])), dtype=np.uint32)
            cur = int(cell[0]) | (int(cell[1]) << 32 if nd == 2 else 0)
            dat = int(w.v[n0, lane]) | (int(w.v[n0 + 1, lane]) << 32 if nd == 2 else 0)
            new = fn(cur, dat) & (M32 if nd == 1 else M64)
            old[lane, 0] = cur & M32
            cell[0] = new & M32
            if nd == 2:
                old[lane, 1] = cur >> 32
                cell[1] = new >> 32
        if ret:
            d0, _ = w.vrange(i.args[0])
            for k in range(nd):
                w.v[d0 + k] = np.where(act, old[:, k], w.v[d0 + k])
    return h


OPS["global_atomic_add"] = _gatomic(lambda c, d: c + d)
OPS["global_atomic_or"] = _gatomic(lambda c, d: c | d)
OPS["global_atomic_and"] = _gatomic(lambda c, d: c & d)
OPS["global_atomic_swap"] = _gatomic(lambda c, d: d)
OPS["global_atomic_umax"] = _gatomic(lambda c, d: max(c, d))
@op("buffer_wbl2", "buffer_inv")
def _(w, i):
    """cache write-back / invalidate of release and acquire at agent scope: the interpreter's memory is one coherent array, nothing to do
    (not among the product's instructions -- its look-back orders with sc1 accesses; here for tools/fuzz_interpreter_vs_compiler.py)"""


OPS["global_atomic_umin"] = _gatomic(lambda c, d: min(c, d))
OPS["global_atomic_xor"] = _gatomic(lambda c, d: c ^ d)
OPS["global_atomic_sub"] = _gatomic(lambda c, d: c - d)
OPS["global_atomic_smax"] = _gatomic(lambda c, d: c if _sx(c, 32) >= _sx(d, 32) else d)
OPS["global_atomic_smin"] = _gatomic(lambda c, d: c if _sx(c, 32) <= _sx(d, 32) else d)
OPS["global_atomic_add_x2"] = _gatomic(lambda c, d: c + d, 2)
OPS["global_atomic_swap_x2"] = _gatomic(lambda c, d: d, 2)


# ------------------------------------------------------------------------------------------------------------------------------
# running a grid
# ------------------------------------------------------------------------------------------------------------------------------

# The gfx9 "manually inserted wait states" that matter around hand-written assembly (the compiler pads its own code; it does not look
# inside an asm statement): a DPP instruction needs 2 wait states after a VALU write of the VGPR it permutes and 5 after a VALU write
# of EXEC; v_readlane / v_writelane need 4 after a VALU write of the SGPR (or VCC) that selects the lane; a vector-memory
# instruction needs 5 after a VALU write (v_readfirstlane, v_readlane, v_cmp ...) of an SGPR it uses as base; and on gfx940 / gfx950
# (LLVM: hasVDecCoExecHazard) a VALU instruction needs 2 after a VALU write of an SGPR or VCC it reads -- the carry between v_add_co
# and v_addc, the mask between v_cmp and v_cndmask (hipcc pads its own: v_sub_co / s_nop 1 / v_subb_co).  One issued instruction
# = one wait state, s_nop N = N + 1.  Checked on the EXECUTED instruction stream, per wavefront; compiler-scheduled code must come
# out clean too (it does: that calibrates the rules).
HAZARD_LOG = []
PROFILE = None  # tools/dynamic_profile.py: a collections.Counter of (kernel, opcode) -> wave-instructions EXECUTED


def _regs_of(tok, kind):
    m = re.match(rf"^{kind}(\d+)$", tok)
    if m:
        return [int(m.group(1))]
    m = re.match(rf"^{kind}\[(\d+):(\d+)\]$", tok)
    if m:
        return list(range(int(m.group(1)), int(m.group(2)) + 1))
    return []


def _hazard_info(ins):
    """(DPP source VGPRs | None, lane-select SGPRs, VMEM SGPRs, VGPRs written, SGPRs written, writes EXEC, wait states issued)"""
    op, args = ins.op, ins.args
    is_valu = op.startswith("v_")
    dpp = _regs_of(args[1], "v") if is_valu and (op.endswith("_dpp") or any(k in ins.mods for k in _DPP_KEYS)) else None
    lane = []
    if op in ("v_readlane_b32", "v_writelane_b32"):
        lane = _regs_of(args[2], "s") or ([106] if args[2] in ("vcc_lo", "vcc") else [])
    vmem = [r for tok in args for r in _regs_of(tok, "s")] if op.startswith(("global_", "buffer_", "flat_", "scratch_")) else []
    vw, sw, dests = [], [], []
    if is_valu and args:
        vw = _regs_of(args[0], "v")
        dests = [args[0]] + ([args[1]] if ("_co_" in op or op.startswith("v_mad_u64")) and len(args) > 1 else [])
        for tok in dests:
            sw += _regs_of(tok, "s")
            if tok in ("vcc", "vcc_lo"):
                sw += [106, 107]
    states = int(args[0], 0) + 1 if op == "s_nop" and args else 1
    sread = []  # SGPRs / VCC a VALU instruction reads as a source (carry-in, select mask, scalar operand)
    if is_valu:
        for tok in args[len(dests) if "_co_" in op or op.startswith(("v_mad_u64", "v_cmp")) else 1:]:
            sread += _regs_of(tok, "s")
            if tok in ("vcc", "vcc_lo"):
                sread += [106, 107]
    return dpp, lane, vmem, vw, sw, is_valu and op.startswith("v_cmpx"), states, sread


def check_hazards(w, ins):
    info = ins.target if isinstance(ins.target, tuple) else None
    if info is None:
        info = _hazard_info(ins)
        if ins.target is None:  # (branches keep their target there; everything else caches its hazard facts in the free slot)
            ins.target = info
    dpp, lane, vmem, vw, sw, wexec, states, sread = info
    now = w.clock
    for r in sread:
        if now - w.sgpr_written.get(r, -100) < 2:
            HAZARD_LOG.append((ins.addr, ins.text, f"VALU reads {'vcc' if r >= 106 else 's%d' % r}, written by VALU {now - w.sgpr_written[r]} wait states ago (2 needed on gfx940 / gfx950)"))
            break
    if dpp is not None:
        for n in dpp:
            if now - w.vgpr_written.get(n, -100) < 2:
                HAZARD_LOG.append((ins.addr, ins.text, f"DPP reads v{n} {now - w.vgpr_written[n]} wait states after a VALU write (2 needed)"))
        if now - w.exec_written < 5:
            HAZARD_LOG.append((ins.addr, ins.text, f"DPP {now - w.exec_written} wait states after a VALU write of EXEC (5 needed)"))
    for r in lane:
        if now - w.sgpr_written.get(r, -100) < 4:
            HAZARD_LOG.append((ins.addr, ins.text, f"lane select s{r} written by VALU {now - w.sgpr_written[r]} wait states ago (4 needed)"))
    for r in vmem:
        if now - w.sgpr_written.get(r, -100) < 5:
            HAZARD_LOG.append((ins.addr, ins.text, f"VMEM uses s{r}, written by VALU {now - w.sgpr_written[r]} wait states ago (5 needed)"))
    for n in vw:
        w.vgpr_written[n] = now + 1
    for r in sw:
        w.sgpr_written[r] = now + 1
    if wexec:
        w.exec_written = now + 1
    w.clock = now + states


# What a missing s_waitcnt would look like.  The interpreter completes every memory operation at once, so a missing wait cannot
# change a result here -- instead the counters are kept as bookkeeping: every vector-memory operation joins the wavefront's vmcnt queue,
# every LDS / scalar-memory operation its lgkmcnt queue, s_waitcnt retires what the counter value guarantees (gfx9: vector-memory
# operations and LDS operations complete in issue order; scalar loads in any order, so with one queued only lgkmcnt(0) says anything),
# and an instruction that reads or writes a register some queued load has yet to deliver, or a barrier crossed with this wavefront's
# LDS stores still queued, is reported.  This is what stands behind the inline assembly's explicit waits (lds_append_complete).
WAIT_LOG = []


def _wait_info(ins):
    """(vector registers touched, scalar registers touched (as -1 - n), what it queues: None | ("vm" | "lds" | "ldsw" | "smem", destination registers))"""
    op, args = ins.op, ins.args
    touched = set()
    for tok in args:
        for n in _regs_of(tok, "v"):
            touched.add(n)
        for n in _regs_of(tok, "s"):
            touched.add(-1 - n)
    queues = None
    if op.startswith("global_load") or (op.startswith("global_atomic") and len(args) == 4):
        queues = ("vm", set(_regs_of(args[0], "v")))
    elif op.startswith(("global_store", "global_atomic")):
        queues = ("vm", set())
    elif op.startswith(("ds_read", "ds_bpermute")):
        queues = ("lds", set(_regs_of(args[0], "v")))
    elif op.startswith("ds_write"):
        queues = ("ldsw", set())
    elif op.startswith("s_load"):
        queues = ("smem", {-1 - n for n in _regs_of(args[0], "s")})
    return touched, queues


def _retire(w, entry):
    for r in entry:
        c = w.pending.get(r, 0) - 1
        if c <= 0:
            w.pending.pop(r, None)
        else:
            w.pending[r] = c


def check_waits(w, ins):
    op = ins.op
    if op == "s_waitcnt":
        if "vmcnt" in ins.mods:
            n = int(ins.mods["vmcnt"], 0)
            while len(w.vm_queue) > n:
                _retire(w, w.vm_queue.pop(0))
        if "lgkmcnt" in ins.mods:
            n = int(ins.mods["lgkmcnt"], 0)
            if n == 0 or not any(k == "smem" for k, _ in w.lgkm_queue):
                while len(w.lgkm_queue) > n:
                    _retire(w, w.lgkm_queue.pop(0)[1])
        return
    if op == "s_barrier":
        if any(k == "ldsw" for k, _ in w.lgkm_queue):
            WAIT_LOG.append((ins.addr, ins.text, "s_barrier with LDS stores of this wavefront not waited for (lgkmcnt)"))
        return
    info = ins.mods.get("_wait")
    if info is None:
        info = ins.mods["_wait"] = _wait_info(ins)
    touched, queues = info
    if w.pending:
        dst = queues[1] if queues else ()
        for r in touched:
            if r in w.pending and r not in dst:  # (a later load into the same register returns behind the earlier one: in order)
                name = f"v{r}" if r >= 0 else f"s{-1 - r}"
                WAIT_LOG.append((ins.addr, ins.text, f"{name} is the destination of a memory operation that has not been waited for"))
    if queues:
        kind, dst = queues
        if kind == "vm":
            w.vm_queue.append(dst)
        else:
            w.lgkm_queue.append((kind, dst))
        for r in dst:
            w.pending[r] = w.pending.get(r, 0) + 1


class Kernel:
    def __init__(self, co: CodeObject, name: str):
        self.co, self.name = co, name
        self.meta = co.kernels[name]
        self.desc = co.descriptor(name)
        if self.desc["preload"] & 0x7F:
            raise Unsupported("kernel-argument preloading")
        if self.meta.get(".private_segment_fixed_size", 0):
            raise Unsupported(f"{name} uses {self.meta['.private_segment_fixed_size']} bytes of scratch per lane")
        ins = co.instructions()
        # the kernel's own instructions: from its entry to the end of its symbol
        _, size = co.symbols[name]
        addrs = [a for a in ins if self.desc["entry"] <= a < self.desc["entry"] + size]
        self.code = {a: ins[a] for a in addrs}
        for a in addrs:
            x = ins[a]
            if x.fn is None:
                x.fn = OPS.get(x.op)
                if x.op.startswith(("s_branch", "s_cbranch")):
                    x.target = a + x.size + 4 * _sx(int(x.args[0], 0), 16)
        self.missing = sorted({x.op for x in self.code.values() if x.fn is None})


def run_grid(kernel: Kernel, grid: int, block: int, dynamic_lds: int, explicit_args: bytes, resident: int = 8, quantum=4000,
             max_instructions: int = 400_000_000, trace=None, order: str = None):
    """Execute `grid` workgroups of `block` work-items.  At most `resident` workgroups are in flight at a time and are started in
    blockIdx order; their wavefronts are interleaved round-robin, `quantum` instructions at a time (and at every s_sleep /
    s_barrier), so spin loops make progress.  (A persistent compress grid with sixteen ticket classes must be resident in full --
    the launcher sizes it that way; `resident` below the grid is for the non-persistent kernels and the single-class case.)
    `order` (default: the environment's GFX950_EXEC_ORDER, else "forward"): the order the workgroups in flight and the wavefronts of
    a workgroup take their turns in -- "forward" (index order: a later wavefront sees what the earlier ones did in their turn),
    "reverse", or "random:<seed>" (reshuffled every pass).  A race only shows when the schedule lets the loser run first: a load the
    compiler sank past a barrier (tools/audit_machine_sink.py) is overtaken by LATER wavefronts' stores only under "reverse"."""
    order = order or os.environ.get("GFX950_EXEC_ORDER", "forward")
    order_rng = random.Random(int(order.split(":")[1])) if order.startswith("random:") else None
    if order not in ("forward", "reverse") and order_rng is None:
        raise ValueError(f"order {order!r}")

    def turns(seq):
        if order == "forward":
            return list(seq)
        if order == "reverse":
            return list(seq)[::-1]
        out = list(seq)
        order_rng.shuffle(out)
        return out

    if kernel.missing:
        raise Unsupported(f"{kernel.name}: no semantics for {kernel.missing}")
    meta, desc = kernel.meta, kernel.desc
    # kernel arguments: the explicit ones as packed by the caller, the hidden ones from the metadata
    karg = np.zeros(max(meta[".kernarg_segment_size"], len(explicit_args)) + 64, dtype=np.uint8)
    karg[:len(explicit_args)] = np.frombuffer(explicit_args, dtype=np.uint8)
    hidden = {"hidden_block_count_x": (grid, 4), "hidden_block_count_y": (1, 4), "hidden_block_count_z": (1, 4), "hidden_group_size_x": (block, 2),
              "hidden_group_size_y": (1, 2), "hidden_group_size_z": (1, 2), "hidden_grid_dims": (1, 2), "hidden_dynamic_lds_size": (dynamic_lds, 4)}
    first_hidden = None
    for a in meta[".args"]:
        kind = a[".value_kind"]
        if kind.startswith("hidden_") and first_hidden is None:
            first_hidden = a[".offset"]
        if kind in hidden:
            val, size = hidden[kind]
            karg[a[".offset"]:a[".offset"] + size] = np.frombuffer(int(val).to_bytes(size, "little"), dtype=np.uint8)
    if first_hidden is not None and len(explicit_args) > first_hidden:
        raise Unsupported(f"{len(explicit_args)} bytes of explicit arguments, but the hidden ones start at {first_hidden}")
    karg_addr = karg.ctypes.data
    lds_bytes = desc["lds_fixed"] + dynamic_lds
    nwaves = (block + 63) // 64  # (the last wavefront of a block that is no multiple of 64 starts with a partial EXEC mask)
    sgpr = 0
    layout = []
    props = desc["props"]
    for bit, n, what in ((0, 4, "private_segment_buffer"), (1, 2, "dispatch_ptr"), (2, 2, "queue_ptr"), (3, 2, "kernarg"), (4, 2, "dispatch_id"),
                         (5, 2, "flat_scratch_init"), (6, 1, "private_segment_size")):
        if (props >> bit) & 1:
            layout.append((what, sgpr, n))
            sgpr += n
    if sgpr != desc["user_sgprs"]:
        raise Unsupported(f"user SGPR layout: {layout} vs USER_SGPR_COUNT {desc['user_sgprs']}")
    if any(w_ not in ("kernarg",) for w_, _, _ in layout):
        raise Unsupported(f"user SGPRs {layout}: only the kernarg segment pointer is provided")

    def make_wg(index):
        wg = Workgroup(index, nwaves, lds_bytes)
        for k in range(nwaves):
            w = Wave(wg, k, kernel.code, desc["entry"], kernel.name)
            for what, at, n in layout:
                w.s[at], w.s[at + 1] = karg_addr & M32, karg_addr >> 32
            s = desc["user_sgprs"]
            if desc["wg_id_x"]:
                w.s[s] = index
                s += 1
            if desc["wg_id_y"]:
                w.s[s] = 0
                s += 1
            if desc["wg_id_z"]:
                w.s[s] = 0
                s += 1
            w.v[0] = (np.arange(64, dtype=np.uint32) + np.uint32(64 * k))  # packed work-item id: x in bits 0-9, y = z = 0
            w.exec = M64 if block - 64 * k >= 64 else (1 << (block - 64 * k)) - 1
            wg.waves.append(w)
        return wg

    next_wg, live, executed = 0, [], 0
    while next_wg < grid or live:
        while next_wg < grid and len(live) < resident:
            live.append(make_wg(next_wg))
            next_wg += 1
        progressed = False
        for wg in turns(live):
            for w in turns(wg.waves):
                if w.state == Wave.DONE or w.state == Wave.BARRIER:
                    continue
                w.state = Wave.RUNNING
                n = 0
                code = w.code
                q = quantum(wg.index) if callable(quantum) else quantum  # (a function of the workgroup: adversarial schedules)
                while w.state == Wave.RUNNING and n < q:
                    ins = code[w.pc]
                    w.pc += ins.size
                    if trace is not None:
                        trace(w, ins)
                    check_hazards(w, ins)
                    check_waits(w, ins)
                    if PROFILE is not None:
                        PROFILE[(kernel.name, ins.op)] += 1
                    try:
                        ins.fn(w, ins)
                    except Unsupported:
                        raise
                    except Exception as e:
                        raise RuntimeError(f"{kernel.name}: workgroup {wg.index} wave {w.index} at {ins.addr:#x}: {ins.text}: {type(e).__name__}: {e}") from e
                    n += 1
                executed += n
                progressed = progressed or n > 0
                if w.state == Wave.DONE:
                    wg.live -= 1
            # barrier: released when every wavefront of the workgroup that is still running has arrived
            waiting = [w for w in wg.waves if w.state == Wave.BARRIER]
            if waiting and len(waiting) == wg.live:
                for w in waiting:
                    w.state = Wave.RUNNING
                progressed = True
            elif waiting and any(w.state == Wave.DONE for w in wg.waves) and len(waiting) == wg.live:
                pass
            if wg.live == 0:
                live.remove(wg)
        if executed > max_instructions:
            raise RuntimeError(f"{kernel.name}: {executed} instructions executed: a hang?")
        if not progressed and live:
            raise RuntimeError(f"{kernel.name}: deadlock: wavefronts wait at a barrier the others never reach")
    return executed


# ------------------------------------------------------------------------------------------------------------------------------
# the bridge: the functional model's host side launches, this interpreter executes
# ------------------------------------------------------------------------------------------------------------------------------

class _DlInfo(C.Structure):
    _fields_ = [("dli_fname", C.c_char_p), ("dli_fbase", C.c_void_p), ("dli_sname", C.c_char_p), ("dli_saddr", C.c_void_p)]


class Bridge:
    """with Bridge(model_library_path, [product .so, stages .so], workdir): every kernel the model's host code launches runs as
    gfx950 code.  `only` restricts that to kernels whose name contains one of the given substrings (the others run on the model)."""

    def __init__(self, model_lib_path: str, hip_libs, workdir: str, only=None, resident: int = 16, trace=None):
        self.model_lib_path = model_lib_path
        self.cos = []
        for n, lib in enumerate(hip_libs):
            self.cos += code_objects_of(lib, os.path.join(workdir, f"lib{n}"))
        self.only, self.resident, self.trace = only, resident, trace
        self.kernels = {}
        self.launched = []  # (kernel name, grid, block, instructions executed) since the last __enter__
        self.total_launches = self.total_instructions = 0
        self.error = None
        # host function address -> mangled name, from the model library's symbol table (local symbols included)
        out = subprocess.run(["nm", "--defined-only", model_lib_path], capture_output=True, text=True, check=True).stdout
        self.host_syms = {}
        for line in out.splitlines():
            p = line.split()
            if len(p) == 3 and p[1] in "tTwW":
                self.host_syms[int(p[0], 16)] = p[2]
        self._cb = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint, C.c_uint, C.c_uint, C.c_void_p, C.c_uint)(self._hook)

    def kernel_named(self, host_name: str):
        k = self.kernels.get(host_name)
        if k is None:
            for co in self.cos:
                if host_name in co.kernels:
                    k = self.kernels[host_name] = Kernel(co, host_name)
                    break
        return k

    def _hook(self, fn, grid, block, lds, args, nbytes):
        if self.error is not None:
            return 0
        try:
            info = _DlInfo()
            if not C.CDLL(None).dladdr(C.c_void_p(fn), C.byref(info)):
                raise Unsupported("dladdr cannot place the kernel's host function")
            name = self.host_syms.get(fn - info.dli_fbase)
            if name is None:
                raise Unsupported(f"no symbol at {fn - info.dli_fbase:#x} of {self.model_lib_path}")
            if self.only is not None and not any(s in name for s in self.only):
                return 0
            k = self.kernel_named(name)
            if k is None:
                raise Unsupported(f"kernel {name} is in none of the gfx950 code objects")
            n = run_grid(k, grid, block, lds, C.string_at(args, nbytes), resident=self.resident, trace=self.trace)
            self.launched.append((name, grid, block, n))
            self.total_launches += 1
            self.total_instructions += n
            return 1
        except BaseException as e:  # (an exception cannot cross the C frames of the model: keep it, let the model run the launch)
            self.error = e
            return 0

    def __enter__(self):
        self.launched, self.error = [], None
        self.lib = C.CDLL(self.model_lib_path)
        self.lib.wavesim_set_launch_hook.argtypes = [C.c_void_p]
        self.lib.wavesim_set_launch_hook(C.cast(self._cb, C.c_void_p))
        return self

    def __exit__(self, *exc):
        self.lib.wavesim_set_launch_hook(None)
        if self.error is not None and exc[0] is None:
            raise self.error
        return False
