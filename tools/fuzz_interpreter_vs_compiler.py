#!/usr/bin/env python3
"""Differential fuzz of the instruction-level interpreter (tests/gfx950_exec.py) against the compiler.

Every offline statement of the form "the BUILT gfx950 code objects reproduce the oracle" rests on the interpreter's reading of the
ISA.  tests/test_interpreter_vs_compiler.py holds it against a handful of hand-written kernels; this tool widens that to RANDOM
programs: each case generates one HIP kernel out of language-defined building blocks --

  * 32- and 64-bit integer arithmetic, shifts, rotates, comparisons and selects, signed forms, widening multiplies,
    popcount / clz / ctz / bit reverse / byte swap, conversions,
  * wavefront crossings (__shfl, __shfl_xor, __shfl_up, __shfl_down with widths, __ballot, __any / __all),
  * LDS exchanges through __syncthreads with 4-, 8- and 16-byte accesses at permuted addresses,
  * divergent if / else regions and data-dependent loops with early exits; ballots and votes INSIDE divergent regions (only the
    lanes that took the branch take part); wave-uniform loops that run until no lane of the wavefront has work left,
  * global loads of 4, 8 and 16 bytes; global atomics (add / or / and / xor / max / min) on a small table, results unused -- what
    LLVM's atomic optimizer turns into a wave reduction and one atomic per wavefront,
  * half of the programs end in a ticket loop (tiles drawn from an atomic counter, broadcast through LDS, barriers inside a
    wave-uniform loop: the product's persistent-kernel shape), a third in a chain across the workgroups (wait for the predecessor's
    flag, add, publish -- release / acquire at agent scope, or flag and sum in one 8-byte relaxed atomic as the product's look-back has it),
  * with --intrinsics, the gfx950 builtins the product's kernels lean on (perm, alignbit, alignbyte, ubfe / sbfe, mbcnt, ds_bpermute,
    ds_permute, readlane, readfirstlane, and DPP moves -- quad_perm, row_shl / shr / ror, wave_shl / shr / rol / ror, row_mirror,
    row_half_mirror, row_bcast:15 / 31, with row / bank masks and bound_ctrl, alone and where the compiler's DPP combiner folds them
    into v_add_u32_dpp / v_and_b32_dpp / ...), whose host meaning is written here from the ISA manual's pseudo-code --

compiles it TWICE: by hipcc for gfx950 (executed by the interpreter, with its hazard and s_waitcnt checkers on -- compiler output
must never trip them) and, restated over arrays of all work-items, by clang++ for the host (executed natively).  The two results
must be the same words.  Kernels in which the compiler used an instruction the interpreter does not know are counted and named,
not failed; a wrong answer is first put to tools/audit_machine_sink.py (this image's LLVM can sink an LDS load past a barrier: such
a case is the compiler's race, reported as "compiler-sunk-load") and then recompiled through the compiler's other instruction
selector (GlobalISel): if that build runs to the host's answer it is reported as "codegen-disagreement", for a human to read;
what is still wrong then is held against tools/audit_bitop3.py (the v_bitop3 truth-table defect both selectors share): "compiler-bitop3".

Further legs: --model (the same device text on the functional model tests/wavesim, under a random fiber schedule) and, on a GPU box,
--hardware (the code object on the device: its words must be the interpreter's).  The interpreter's schedule varies with the case
(index order / reverse / reshuffled every pass; 60 - 2 500 instructions per turn).

    python3 tools/fuzz_interpreter_vs_compiler.py --seed 1 --cases 100 [--opt O1|O2|O3|Os] [--intrinsics] [--model] [--hardware] [--keep DIR]
"""
import argparse
import collections
import ctypes as C
import os
import random
import struct
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
HIPCC = "/opt/rocm/bin/hipcc"
CLANG = "/opt/rocm/lib/llvm/bin/clang++"
ATOMIC_BASE = {"add": 0, "or": 16, "max": 32, "min": 48, "and": 64, "xor": 80}   # 16 slots per operation in the `table` argument
TABLE_WORDS = 104
PAIR_SLOT = 100                # (8-byte aligned: the "packed" chain's (sum, flag) pairs)
FLAG_SLOT, SUM_SLOT = 97, 99   # the chain: workgroup b waits for workgroup b - 1's flag, adds its word to that one's sum, publishes both (look-back)
TICKET_SLOT, TILES = 96, 5   # the ticket loop: workgroups draw tile numbers from table[96] until they run out (the product's persistent-kernel shape)
BLOCK, GRID = 256, 2
N = BLOCK * GRID
NIN = 6  # input planes of N words


class Gen:
    """One random program.  A statement is (kind, payload); expressions are format strings over {name} placeholders so that the
    device text (scalars) and the host text (arrays indexed by [i]) are the same characters."""

    def __init__(self, rng: random.Random, intrinsics: bool = False, model: bool = False):
        self.r = rng
        self.intrinsics = intrinsics
        self.model = model      # also run on the functional model (tests/wavesim): leave out what it has no primitive for (ds_permute)
        self.v32 = ["a", "b", "c", "d"]
        self.v64 = ["p", "q"]
        self.n = 0
        self.stmts = []

    def new(self, wide: bool) -> str:
        self.n += 1
        name = f"{'w' if wide else 'v'}{self.n}"
        (self.v64 if wide else self.v32).append(name)
        return name

    def x32(self) -> str:
        r = self.r
        k = r.random()
        if k < 0.75:
            return "{" + r.choice(self.v32) + "}"
        if k < 0.9:
            return f"{r.choice([0, 1, 2, 3, 5, 31, 32, 63, 64, 255, 256, 0xffff, 0x80000000, 0xffffffff, r.getrandbits(32)]):#x}u"
        return "(uint32_t) ({" + r.choice(self.v64) + "}" + r.choice(["", " >> 32", " >> 17"]) + ")"

    def x64(self) -> str:
        r = self.r
        k = r.random()
        if k < 0.7:
            return "{" + r.choice(self.v64) + "}"
        if k < 0.85:
            return f"{r.choice([0, 1, 0xffffffff, 0x100000000, 0x8000000000000000, 0xffffffffffffffff, r.getrandbits(64)]):#x}ull"
        return "(uint64_t) " + self.x32()

    def e32(self) -> str:
        r = self.r
        x, y, z = self.x32(), self.x32(), self.x32()
        k = r.randrange(1, 32)
        forms = [
            f"{x} + {y}", f"{x} - {y}", f"{x} * {y}", f"{x} ^ {y}", f"{x} & {y}", f"{x} | {y}", f"~{x}", f"{x} & ~{y}",
            f"{x} << ({y} & 31u)", f"{x} >> ({y} & 31u)", f"{x} << {k}u", f"{x} >> {k}u",
            f"({x} << {k}u) | ({x} >> {32 - k}u)", f"({x} << {k}u) | ({y} >> {32 - k}u)",
            f"(uint32_t) ((int32_t) {x} >> {k})", f"(uint32_t) ((int32_t) {x} >> ({y} & 31u))",
            f"({x} < {y}) ? {z} : {x}", f"((int32_t) {x} < (int32_t) {y}) ? {y} : {z}", f"({x} == {y}) ? {z} : {y}",
            f"({x} <= {y}) ? {z} : {x}", f"((int32_t) {x} <= (int32_t) {y}) ? {y} : {z}", f"({x} != {y}) ? {z} : {y}",
            f"((int32_t) {x} >= (int32_t) {y}) ? {x} : {z}", f"({x} >= {y}) ? {y} : {z}",
            f"({x} < {y} ? {x} : {y})", f"({x} > {y} ? {x} : {y})", f"((int32_t) {x} > (int32_t) {y} ? {x} : {y})",
            f"(uint32_t) __builtin_popcount({x})", f"(uint32_t) __builtin_clz({x} | 1u)", f"(uint32_t) __builtin_ctz({x} | 0x80000000u)",
            f"__builtin_bitreverse32({x})", f"__builtin_bswap32({x})",
            f"({x} >> {k % 24}u) & {(1 << r.randrange(1, 9)) - 1:#x}u", f"({x} & {r.getrandbits(32):#x}u) | ({y} & ~{r.getrandbits(32):#x}u)",
            f"(uint32_t) (((uint64_t) {x} * {y}) >> 32)", f"(uint32_t) (((int64_t) (int32_t) {x} * (int64_t) (int32_t) {y}) >> 32)",
            f"{x} * {r.choice([3, 5, 9, 17, 0x9e3779b9])}u + {y}", f"({x} + {y}) ^ ({x} >> 1)", f"({x} ^ {y}) & ({y} ^ {z})",
            f"({x} & {y}) | (~{x} & {z})", f"({x} != 0u) + ({y} != 0u) + ({z} > 7u)",
            f"(uint32_t) (uint8_t) ({x} >> 8) + ({y} >> 24)",
            f"(uint32_t) (int32_t) (int16_t) {x}", f"(uint32_t) (int32_t) (int8_t) ({x} >> {r.choice([0, 8, 16])}u)",
            f"{x} / ({y} | 1u)", f"{x} % ({y} | 1u)",
        ]
        # byte and halfword shuffles of two words (what the compiler's v_perm_b32 / SDWA / v_alignbyte matchers feed on)
        def pick(j):
            src, i = r.choice([x, y]), r.randrange(4)
            return f"((({src} >> {8 * i}u) & 0xffu) << {8 * j}u)"
        forms += [" | ".join(pick(j) for j in range(4)), " | ".join(pick(j) for j in r.sample(range(4), 3)),
                  f"({x} << 16) | ({y} >> 16)", f"({x} & 0xffff0000u) | ({y} & 0xffffu)", f"({x} >> 16) | ({y} & 0xffff0000u)",
                  f"(({x} & 0x00ff00ffu) << 8) | (({x} >> 8) & 0x00ff00ffu)", f"({x} & 0xff00ff00u) | (({y} >> 8) & 0x00ff00ffu)"]
        # nibble, bit-pair and single-bit interleaves (the inner steps of the product's 32 x 32 / 64 x 64 bit transposes)
        forms += [f"(({x} >> 4) & 0x0f0f0f0fu) | (({y} << 4) & 0xf0f0f0f0u)", f"({x} & 0x0f0f0f0fu) | (({y} & 0x0f0f0f0fu) << 4)",
                  f"(({x} >> 2) & 0x33333333u) | ({y} & 0xccccccccu)", f"(({x} >> 1) & 0x55555555u) | (({y} << 1) & 0xaaaaaaaau)",
                  f"__builtin_amdgcn_perm({x}, {y}, 0x07030602u) >> 4" if self.intrinsics else f"({x} >> 4) ^ ({y} << 28)",
                  f"((__builtin_amdgcn_perm({x}, {y}, 0x05010400u) >> 4) & 0x0f0f0f0fu) | (__builtin_amdgcn_perm({x}, {y}, 0x05010400u) & 0xf0f0f0f0u)" if self.intrinsics else f"({x} << 4) ^ ({y} >> 28)"]
        if self.intrinsics:  # gfx950 builtins the product's kernels lean on; their host meaning is HOST_PRELUDE's (from the ISA manual)
            sel = sum(r.choice([0, 1, 2, 3, 4, 5, 6, 7, 0x0c]) << (8 * j) for j in range(4))
            forms += [
                f"__builtin_amdgcn_perm({x}, {y}, {sel:#x}u)", f"__builtin_amdgcn_perm({x}, {y}, {sel:#x}u)",
                f"__builtin_amdgcn_alignbit({x}, {y}, {z})", f"__builtin_amdgcn_alignbit({x}, {y}, {k}u)", f"__builtin_amdgcn_alignbyte({x}, {y}, {z})",
                f"__builtin_amdgcn_ubfe({x}, {y}, {z})", f"__builtin_amdgcn_ubfe({x}, {k}u, {r.randrange(0, 32)}u)",
                f"(uint32_t) __builtin_amdgcn_sbfe((int32_t) {x}, {k}u, {r.randrange(0, 32)}u)",
                f"__builtin_amdgcn_mbcnt_hi({y}, __builtin_amdgcn_mbcnt_lo({x}, {z}))", f"__builtin_amdgcn_mbcnt_lo({x}, 0u)",
            ]
        return r.choice(forms)

    def e64(self) -> str:
        r = self.r
        x, y = self.x64(), self.x64()
        s = self.x32()
        k = r.randrange(1, 64)
        forms = [
            f"{x} + {y}", f"{x} - {y}", f"{x} ^ {y}", f"{x} & {y}", f"{x} | {y}", f"~{x}", f"{x} * {y}",
            f"{x} << ({s} & 63u)", f"{x} >> ({s} & 63u)", f"{x} << {k}u", f"{x} >> {k}u", f"({x} << {k}u) | ({x} >> {64 - k}u)",
            f"(uint64_t) ((int64_t) {x} >> {k})", f"({x} < {y}) ? {x} : {y}", f"((int64_t) {x} < (int64_t) {y}) ? {y} : {x}",
            f"({x} <= {y}) ? {y} : {x}", f"((int64_t) {x} >= (int64_t) {y}) ? {y} : {x}", f"({x} != {y}) ? {x} : ~{y}", f"((int64_t) {x} <= (int64_t) {y}) ? {x} : {y}",
            f"(uint64_t) {s} * (uint64_t) {self.x32()}", f"{x} + (uint64_t) {s}", f"({x} << 32) | (uint64_t) {s}",
            f"(uint64_t) __builtin_popcountll({x})", f"(uint64_t) __builtin_clzll({x} | 1ull)", f"(uint64_t) __builtin_ctzll({x} | 0x8000000000000000ull)",
            f"(uint64_t) (int64_t) (int32_t) {s}", f"{x} * 0x9e3779b97f4a7c15ull + {y}", f"({x} == {y}) ? 1ull : ({x} ^ {y})",
            f"(({x} >> 4) & 0x0f0f0f0f0f0f0f0full) | (({y} << 4) & 0xf0f0f0f0f0f0f0f0ull)", f"(({x} >> 32) | ({y} << 32))", f"(({x} & 0xffffffff00000000ull) | ({y} >> 32))",
            f"(({x} >> 8) & 0x00ff00ff00ff00ffull) | (({y} << 8) & 0xff00ff00ff00ff00ull)", f"(({x} >> 16) & 0x0000ffff0000ffffull) | (({y} << 16) & 0xffff0000ffff0000ull)",
        ]
        return r.choice(forms)

    def plain(self, depth=0):
        """a work-item-local statement: ('set', name, wide, expr)"""
        if self.r.random() < 0.3:
            e = self.e64()                      # (the expression first: a new name must not appear in its own initialiser)
            return ("set", self.new(True), True, e)
        e = self.e32()
        return ("set", self.new(False), False, e)

    def assign_existing(self):
        wide = self.r.random() < 0.25
        name = self.r.choice((self.v64 if wide else self.v32)[2 if wide else 4:] or (self.v64 if wide else self.v32))
        return ("upd", name, wide, self.e64() if wide else self.e32())

    def statement(self):
        r = self.r
        k = r.random()
        if k < 0.04:
            # a butterfly step of a bit transpose (the product's transposes are chains of these): t = ((x >> k) ^ y) & m; y ^= t; x ^= t << k
            wide = r.random() < 0.4
            pool = self.v64 if wide else self.v32
            x, y = r.choice(pool), r.choice(pool)
            kk, m = r.choice([(1, 0x5555555555555555), (2, 0x3333333333333333), (4, 0x0f0f0f0f0f0f0f0f), (8, 0x00ff00ff00ff00ff), (16, 0x0000ffff0000ffff)] + ([(32, 0xffffffff)] if wide else []))
            m &= (1 << (64 if wide else 32)) - 1
            suf = "ull" if wide else "u"
            t = self.new(wide)
            y2 = self.new(wide)
            x2 = self.new(wide)
            return ("multi", [("set", t, wide, f"(({{{x}}} >> {kk}u) ^ {{{y}}}) & {m:#x}{suf}"), ("set", y2, wide, f"{{{y}}} ^ {{{t}}}"), ("set", x2, wide, f"{{{x}}} ^ ({{{t}}} << {kk}u)")])
        if k < 0.50:
            return self.plain()
        if k < 0.68:
            src = r.choice(self.v32) if r.random() < 0.8 else r.choice(self.v64)
            wide = src in self.v64
            width = r.choice([64, 64, 64, 32, 16, 8])
            how = r.choice(["xor", "up", "down", "idx", "idxv"] + ((["bperm", "readlane", "first"] + ([] if self.model else ["perm"])) if self.intrinsics and not wide else []))
            if how in ("bperm", "perm", "readlane", "first"):
                width = 64
            delta = r.randrange(1, width) if how != "idx" else r.randrange(0, width)
            sel = r.choice(self.v32)
            return ("cross", self.new(wide), wide, src, how, delta, width, sel)
        if k < 0.70 and self.intrinsics:
            # a DPP move, alone or where the compiler's DPP combiner can fold it into the consuming VALU instruction (v_add_u32_dpp ...)
            src, other = r.choice(self.v32), r.choice(self.v32)
            ctrl = r.choice([r.randrange(256), 0x100 + r.randrange(1, 16), 0x110 + r.randrange(1, 16), 0x120 + r.randrange(1, 16),
                             0x130, 0x134, 0x138, 0x13c, 0x140, 0x141, 0x142, 0x143, 0x111, 0x112, 0x114, 0x118, 0x142, 0x143])
            form = r.choice(["plain", "plain", "add", "and", "or", "xor", "masked"])
            rm, bm, bc = (0xf, 0xf, r.random() < 0.5) if form != "masked" else (r.choice([0xf, 0xa, 0xc, 0x5, 0x8]), r.choice([0xf, 0xf, 0x3, 0xe]), r.random() < 0.5)
            return ("dpp", self.new(False), src, other, ctrl, rm, bm, bc, form)
        if k < 0.74:
            src = r.choice(self.v32)
            return ("ballot", self.new(True), src, r.randrange(32))
        if k < 0.78:
            src = r.choice(self.v32)
            return ("vote", self.new(False), src, r.choice(["any", "all"]), r.getrandbits(32) >> r.randrange(20, 32))
        if k < 0.80:
            # atomics on a small table in global memory, results not used: the table's final content does not depend on the order
            how = r.choice(["add", "add", "or", "max", "min", "and", "xor"])
            return ("atomic", how, r.choice(self.v32), r.choice(self.v32), r.choice([1, 3, 7, 15]))
        if k < 0.815:
            # two LDS reads a fixed distance apart (ds_read2 / ds_read2st64 material)
            src = r.choice(self.v64)
            mul_w = r.choice([1, 3, 5, 7, 9, 11, 13, 15, 17, 33, 65, 127, 129, 255])
            return ("ldspair", self.new(True), src, mul_w, r.randrange(256), r.choice([1, 2, 64, 65, 128]))
        if k < 0.88:
            kind = r.choice(["u32", "u32", "u64", "u128"])
            mul_w = r.choice([1, 3, 5, 7, 9, 11, 13, 15, 17, 33, 65, 127, 129, 255])
            add_w, mul_r, add_r = r.randrange(256), r.randrange(1, 64), r.randrange(256)
            if kind == "u32":
                src = r.choice(self.v32)
                return ("lds32", self.new(False), src, mul_w, add_w, mul_r, add_r)
            if kind == "u64":
                src = r.choice(self.v64)
                return ("lds64", self.new(True), src, mul_w, add_w, mul_r, add_r)
            srcs = [r.choice(self.v32) for _ in range(4)]
            return ("lds128", self.new(False), srcs, mul_w, add_w, mul_r, add_r)
        if k < 0.905 and self.model:
            return self.plain()  # (the model refuses wave operations that only part of a wavefront reaches -- by design, loudly)
        if k < 0.905:
            # a ballot / vote INSIDE a divergent region: only the lanes that took the branch take part
            cond = f"({self.x32()} {r.choice(['&', '^', '+'])} {self.x32()}) {r.choice(['& 1u', '& 4u', '> 0x7fffffffu', '% 3u == 1u'])}"
            src = r.choice(self.v32)
            if r.random() < 0.6:
                return ("ifballot", self.new(True), cond, src, r.randrange(32))
            return ("ifvote", self.new(False), cond, src, r.choice(["any", "all"]), r.getrandbits(32) >> r.randrange(1, 8))
        if k < 0.92:
            # a wave-uniform loop: every lane iterates until no lane of the wavefront has work left
            src = r.choice(self.v32)
            return ("waveloop", self.new(False), src, r.choice([1, 2, 3, 5]), r.choice([0xff, 0xfff, 0xffff]))
        if k < 0.95:
            cond = f"({self.x32()} {r.choice(['&', '^', '+'])} {self.x32()}) {r.choice(['& 1u', '& 4u', '> 0x7fffffffu', '< 0x40000000u', '% 3u == 1u'])}"
            then = [self.assign_existing() for _ in range(r.randrange(1, 4))]
            other = [self.assign_existing() for _ in range(r.randrange(0, 3))]
            return ("if", cond, then, other)
        acc = r.choice(self.v32[4:] or self.v32)
        trip = f"({self.x32()} & {r.choice([3, 7, 15])}u)"
        body = r.choice([f"{{{acc}}} * 3u + k", f"{{{acc}}} ^ ({self.x32()} >> (k & 31u))", f"({{{acc}}} << 1) + ({self.x32()} & k)"])
        stop = f"{{{acc}}} > {r.choice([0xf0000000, 0xc0000000, 0xffffff00]):#x}u" if r.random() < 0.5 else None
        return ("loop", acc, trip, body, stop)

    def build(self, nstmts: int):
        self.stmts = [self.statement() for _ in range(nstmts)]
        r = self.r
        # half of the programs end in a ticket loop: (multiplier, offset) of the LDS write and read permutations, a mixing constant
        self.ticket = (r.choice([1, 3, 5, 7, 9, 11, 13, 15, 17, 33, 65, 127, 129, 255]), r.randrange(256), r.randrange(1, 256), r.randrange(256),
                       r.getrandbits(32), r.choice([1, 2, 3])) if r.random() < 0.5 else None
        # ... and a third in a chain across the workgroups: release / acquire at agent scope, a spin with s_sleep (the product's look-back)
        # ("packed": flag and sum in ONE 8-byte relaxed atomic, the product's form; "fenced": two words, release / acquire)
        self.chain = (r.getrandbits(32), r.choice(self.v32), r.choice(["fenced", "packed"])) if r.random() < 0.35 else None
        return self


# The host's meaning of the gfx950 builtins, written from the ISA manual's pseudo-code (V_PERM_B32, V_ALIGNBIT_B32, V_ALIGNBYTE_B32,
# V_BFE_U32 / V_BFE_I32, V_MBCNT_LO / _HI_U32_B32); `lane` is the work-item's position in its wavefront.
HOST_PRELUDE = r"""
#define lane (i & 63u)
static inline uint32_t __builtin_amdgcn_perm(uint32_t s0, uint32_t s1, uint32_t sel) {
    const uint64_t both = ((uint64_t) s0 << 32) | s1; uint32_t d = 0;
    for (int k = 0; k < 4; ++k) { const uint32_t c = (sel >> (8 * k)) & 0xffu;
        uint32_t byte = c <= 7 ? (uint32_t) (both >> (8 * c)) & 0xffu : c == 0x0c ? 0u : 0xffu;
        d |= byte << (8 * k); }
    return d;
}
static inline uint32_t __builtin_amdgcn_alignbit(uint32_t hi, uint32_t lo, uint32_t s) { return (uint32_t) ((((uint64_t) hi << 32) | lo) >> (s & 31u)); }
static inline uint32_t __builtin_amdgcn_alignbyte(uint32_t hi, uint32_t lo, uint32_t s) { return (uint32_t) ((((uint64_t) hi << 32) | lo) >> (8u * (s & 3u))); }
static inline uint32_t __builtin_amdgcn_ubfe(uint32_t v, uint32_t off, uint32_t n) { off &= 31u; n &= 31u; return n ? (v >> off) & ((1u << n) - 1u) : 0u; }
static inline int32_t __builtin_amdgcn_sbfe(int32_t v, uint32_t off, uint32_t n) { off &= 31u; n &= 31u; if (!n) return 0;
    const uint32_t f = (uint32_t) (v >> off) & ((1u << n) - 1u);  /* arithmetic shift: a field past bit 31 is filled with the sign */ return (f >> (n - 1u)) & 1u ? (int32_t) (f | ~((1u << n) - 1u)) : (int32_t) f; }
/* V_MOV_B32_DPP with every lane active (the tool uses it at the top level only): the ISA manual's DPP_CTRL table */
static inline uint32_t dpp_host(uint32_t old, const uint32_t *src, uint32_t l, uint32_t ctrl, uint32_t rm, uint32_t bm, bool bc) {
    const uint32_t row = l >> 4, r = l & 15u; int from = -1;
    if (!((rm >> row) & 1u) || !((bm >> (r >> 2)) & 1u)) return old;
    if (ctrl <= 0xffu) from = (int) ((l & ~3u) | ((ctrl >> (2u * (l & 3u))) & 3u));
    else if (ctrl >= 0x101u && ctrl <= 0x10fu) { const uint32_t s = r + (ctrl - 0x100u); if (s < 16u) from = (int) (row * 16u + s); }
    else if (ctrl >= 0x111u && ctrl <= 0x11fu) { const int s = (int) r - (int) (ctrl - 0x110u); if (s >= 0) from = (int) (row * 16u) + s; }
    else if (ctrl >= 0x121u && ctrl <= 0x12fu) from = (int) (row * 16u + ((r - (ctrl - 0x120u)) & 15u));
    else if (ctrl == 0x130u) { if (l + 1u < 64u) from = (int) l + 1; }
    else if (ctrl == 0x134u) from = (int) ((l + 1u) & 63u);
    else if (ctrl == 0x138u) { if (l >= 1u) from = (int) l - 1; }
    else if (ctrl == 0x13cu) from = (int) ((l - 1u) & 63u);
    else if (ctrl == 0x140u) from = (int) (row * 16u + (15u - r));
    else if (ctrl == 0x141u) from = (int) (row * 16u + ((r & 8u) | (7u - (r & 7u))));
    else if (ctrl == 0x142u) { if (row > 0u) from = (int) ((row - 1u) * 16u + 15u); }
    else if (ctrl == 0x143u) { if (row >= 2u) from = 31; }
    return from < 0 ? (bc ? 0u : old) : src[from];
}
#define __builtin_amdgcn_mbcnt_lo(m, add) ((uint32_t) __builtin_popcount((m) & (lane >= 32u ? 0xffffffffu : (1u << lane) - 1u)) + (add))
#define __builtin_amdgcn_mbcnt_hi(m, add) ((uint32_t) __builtin_popcount((m) & (lane <= 32u ? 0u : (1u << (lane - 32u)) - 1u)) + (add))
"""


def _dpp_text(s, host: bool) -> str:
    """the DPP statement's right-hand side; on the host the builtin is dpp_host(old, source array of the wavefront, lane, ...)"""
    _, name, src, other, ctrl, rm, bm, bc, form = s
    ident = {"plain": None, "masked": None, "add": "0", "or": "0", "xor": "0", "and": "-1"}[form]
    oth = f"{other}[i]" if host else other
    old = oth if ident is None else ident
    if host:
        call = f"dpp_host((uint32_t) {old}, &{src}[w0], l, {ctrl:#x}u, {rm:#x}u, {bm:#x}u, {'true' if bc else 'false'})"
    else:
        call = f"(uint32_t) __builtin_amdgcn_update_dpp((int) {old}, (int) {src}, {ctrl:#x}, {rm:#x}, {bm:#x}, {'true' if bc else 'false'})"
    return call if ident is None else f"{oth} {dict(add='+', xor='^', **{'or': '|', 'and': '&'})[form]} {call}"


def _fmt(expr: str, host: bool) -> str:
    class M(dict):
        def __missing__(self, key):
            return f"{key}[i]" if host else key
    return expr.format_map(M())


def device_source(g: Gen) -> str:
    o = ["#include <hip/hip_runtime.h>", "#include <cstdint>",
         'extern "C" __global__ void __launch_bounds__(256) k_fuzz(const uint32_t *in, uint32_t *out, uint32_t *table) {',
         "    __shared__ uint32_t l32[256]; __shared__ uint64_t l64[256]; __shared__ uint4 l128[256];",
         "    const uint32_t t = threadIdx.x, i = blockIdx.x * 256u + t;",
         f"    uint32_t a = in[i], b = in[i + {N}], c = in[2 * {N} + (i ^ 1u)];",
         f"    const uint2 pq = ((const uint2 *) (in + 3 * {N}))[i >> 1];",
         f"    uint32_t d = (i & 1u) ? pq.y : pq.x;",
         f"    const uint4 four = ((const uint4 *) (in + 4 * {N}))[i >> 2];",
         f"    uint64_t p = ((uint64_t) four.x << 32) | four.w, q = ((uint64_t) four.y << 32) | (four.z ^ in[5 * {N} + i]);"]

    def emit(s, ind):
        pad = "    " * ind
        k = s[0]
        if k == "multi":
            for x in s[1]:
                emit(x, ind)
        elif k == "set":
            o.append(f"{pad}{'uint64_t' if s[2] else 'uint32_t'} {s[1]} = {_fmt(s[3], False)};")
        elif k == "upd":
            o.append(f"{pad}{s[1]} = {_fmt(s[3], False)};")
        elif k == "cross":
            _, name, wide, src, how, delta, width, sel = s
            ty = "uint64_t" if wide else "uint32_t"
            cast = "(unsigned long long) " if wide else ""
            call = {"xor": f"__shfl_xor({cast}{src}, {delta}, {width})", "up": f"__shfl_up({cast}{src}, {delta}u, {width})",
                    "down": f"__shfl_down({cast}{src}, {delta}u, {width})", "idx": f"__shfl({cast}{src}, {delta}, {width})",
                    "idxv": f"__shfl({cast}{src}, (int) (({sel} >> 3) & 63u), {width})",
                    "bperm": f"__builtin_amdgcn_ds_bpermute((int) ({sel} << 2), (int) {src})",
                    "perm": f"__builtin_amdgcn_ds_permute((int) (((t * {2 * delta + 1}u + 5u) & 63u) << 2), (int) {src})",
                    "readlane": f"__builtin_amdgcn_readlane((int) {src}, {delta})",
                    "first": f"__builtin_amdgcn_readfirstlane((int) {src})"}[how]
            o.append(f"{pad}{ty} {name} = ({ty}) {call};")
        elif k == "dpp":
            o.append(f"{pad}uint32_t {s[1]} = {_dpp_text(s, False)};")
        elif k == "ballot":
            o.append(f"{pad}uint64_t {s[1]} = __ballot(({s[2]} >> {s[3]}u) & 1u);")
        elif k == "vote":
            o.append(f"{pad}uint32_t {s[1]} = (uint32_t) __{s[3]}({s[2]} > {s[4]:#x}u);")
        elif k in ("lds32", "lds64"):
            _, name, src, mw, aw, mr, ar = s
            arr, ty = ("l32", "uint32_t") if k == "lds32" else ("l64", "uint64_t")
            o.append(f"{pad}{arr}[(t * {mw}u + {aw}u) & 255u] = {src}; __syncthreads();")
            o.append(f"{pad}{ty} {name} = {arr}[(t * {mr}u + {ar}u) & 255u]; __syncthreads();")
        elif k == "atomic":
            _, how, idx, val, mask = s
            fn = {"add": "atomicAdd", "or": "atomicOr", "max": "atomicMax", "min": "atomicMin", "and": "atomicAnd", "xor": "atomicXor"}[how]
            o.append(f"{pad}{fn}(&table[{ATOMIC_BASE[how]} + (({idx} >> 5) & {mask}u)], {val});")
        elif k == "ldspair":
            _, name, src, mw, aw, dist = s
            o.append(f"{pad}l64[(t * {mw}u + {aw}u) & 255u] = {src}; __syncthreads();")
            o.append(f"{pad}uint64_t {name} = l64[t & 127u] - l64[(t & 127u) + {dist}u]; __syncthreads();")
        elif k == "lds128":
            _, name, srcs, mw, aw, mr, ar = s
            o.append(f"{pad}l128[(t * {mw}u + {aw}u) & 255u] = make_uint4({', '.join(srcs)}); __syncthreads();")
            o.append(f"{pad}const uint4 {name}_q = l128[(t * {mr}u + {ar}u) & 255u]; __syncthreads();")
            o.append(f"{pad}uint32_t {name} = {name}_q.x ^ ({name}_q.y << 1) ^ ({name}_q.z >> 1) ^ ({name}_q.w * 3u);")
        elif k == "ifballot":
            o.append(f"{pad}uint64_t {s[1]} = 0; if ({_fmt(s[2], False)}) {s[1]} = __ballot(({s[3]} >> {s[4]}u) & 1u);")
        elif k == "ifvote":
            o.append(f"{pad}uint32_t {s[1]} = 7u; if ({_fmt(s[2], False)}) {s[1]} = (uint32_t) __{s[4]}({s[3]} > {s[5]:#x}u);")
        elif k == "waveloop":
            o.append(f"{pad}uint32_t {s[1]} = 0; for (uint32_t rest = {s[2]} & {s[4]:#x}u; __any(rest != 0u); rest >>= {s[3]}u) {s[1]} += (rest & 1u) + 1u;")
        elif k == "if":
            o.append(f"{pad}if ({_fmt(s[1], False)}) {{")
            for x in s[2]:
                emit(x, ind + 1)
            if s[3]:
                o.append(f"{pad}}} else {{")
                for x in s[3]:
                    emit(x, ind + 1)
            o.append(f"{pad}}}")
        elif k == "loop":
            _, acc, trip, body, stop = s
            o.append(f"{pad}for (uint32_t k = 0, n = {_fmt(trip, False)}; k < n; ++k) {{")
            o.append(f"{pad}    {acc} = {_fmt(body, False)};")
            if stop:
                o.append(f"{pad}    if ({_fmt(stop, False)}) break;")
            o.append(f"{pad}}}")

    for s in g.stmts:
        emit(s, 1)
    if g.chain and g.chain[2] == "packed":
        mix, var, _ = g.chain
        o += ["    if (t == 0) {   // workgroup b: wait for b - 1's (flag, sum) pair -- one 8-byte relaxed atomic --, add, publish its own",
              f"        unsigned long long *pairs = (unsigned long long *) (table + {PAIR_SLOT});",
              f"        const uint32_t mine = (in[blockIdx.x * 7u + 3u] ^ {mix:#x}u) + ({var} & 0u);",
              "        unsigned long long seen = 1ull << 32;",
              "        if (blockIdx.x > 0) {",
              "            uint32_t polls = 0;   // bounded (a second or so on the device): a predecessor that never publishes is a wrong word, not a hung GPU",
              "            do { seen = __hip_atomic_load(&pairs[blockIdx.x - 1u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); if (!(seen >> 32)) __builtin_amdgcn_s_sleep(1); } while (!(seen >> 32) && ++polls < (1u << 22));",
              "        }",
              "        __hip_atomic_store(&pairs[blockIdx.x], (1ull << 32) | (uint32_t) ((uint32_t) seen + mine), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);",
              "    }"]
    elif g.chain:
        mix, var, _ = g.chain
        o += ["    if (t == 0) {   // workgroup b: wait for b - 1, add, publish -- data first, then the flag with release",
              f"        const uint32_t mine = (in[blockIdx.x * 7u + 3u] ^ {mix:#x}u) + ({var} & 0u);",
              "        uint32_t before = 0;",
              "        if (blockIdx.x > 0) {",
              f"            for (uint32_t polls = 0; polls < (1u << 22) && __hip_atomic_load(&table[{FLAG_SLOT} + blockIdx.x - 1u], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == 0u; ++polls) __builtin_amdgcn_s_sleep(1);   // bounded: see the packed form",
              f"            before = __hip_atomic_load(&table[{SUM_SLOT} + blockIdx.x - 1u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);",
              "        }",
              f"        __hip_atomic_store(&table[{SUM_SLOT} + blockIdx.x], before + mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);",
              f"        __hip_atomic_store(&table[{FLAG_SLOT} + blockIdx.x], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);",
              "    }"]
    if g.ticket:
        mw, aw, mr, ar, mix, reads = g.ticket
        o += ["    for (;;) {   // tiles are handed out by an atomic counter; which workgroup gets which tile does not matter to the result",
              f"        if (t == 0) l32[0] = atomicAdd(&table[{TICKET_SLOT}], 1u);",
              "        __syncthreads();",
              "        const uint32_t tile = l32[0];",
              "        __syncthreads();",
              f"        if (tile >= {TILES}u) break;",
              f"        const uint32_t sv = in[(tile * 97u + t * 3u) % {NIN * N}u] ^ {mix:#x}u;",
              f"        l32[(t * {mw}u + {aw}u) & 255u] = sv; __syncthreads();",
              f"        uint32_t rv = l32[(t * {mr}u + {ar}u) & 255u];"]
        if reads >= 2:
            o.append(f"        rv += l32[(t + 1u) & 255u] >> 3;")
        if reads >= 3:
            o.append(f"        rv ^= l32[(t ^ 32u) & 255u] * 5u;")
        o += ["        __syncthreads();",
              f"        out[(uint64_t) {N}u * STRIDE_ + tile * 256u + t] = rv + tile;",
              "    }"]
    o.append("    uint32_t h = 0x811c9dc5u; uint64_t hh = 0xcbf29ce484222325ull;")
    for idx, v in enumerate(g.v32):
        o.append(f"    h = (h ^ {v}) * 0x01000193u;")
        o.append(f"    out[(uint64_t) i * {len(g.v32) + 2 * len(g.v64) + 3} + {idx}] = {v};")
    base = len(g.v32)
    for idx, v in enumerate(g.v64):
        o.append(f"    hh = (hh ^ {v}) * 0x100000001b3ull;")
        o.append(f"    out[(uint64_t) i * {len(g.v32) + 2 * len(g.v64) + 3} + {base + 2 * idx}] = (uint32_t) {v};")
        o.append(f"    out[(uint64_t) i * {len(g.v32) + 2 * len(g.v64) + 3} + {base + 2 * idx + 1}] = (uint32_t) ({v} >> 32);")
    stride = len(g.v32) + 2 * len(g.v64) + 3
    o.append(f"    out[(uint64_t) i * {stride} + {stride - 3}] = h;")
    o.append(f"    out[(uint64_t) i * {stride} + {stride - 2}] = (uint32_t) hh;")
    o.append(f"    out[(uint64_t) i * {stride} + {stride - 1}] = (uint32_t) (hh >> 32);")
    o.append("}")
    return ("\n".join(o) + "\n").replace("STRIDE_", str(stride))


def host_source(g: Gen) -> str:
    """The same program over arrays of all N work-items: work-item-local statements loop over i; a crossing loops over the 64
    lanes of each wavefront; an LDS exchange is a scatter then a gather per workgroup."""
    o = ["#include <cstdint>", "#include <vector>", f"static const uint32_t N = {N};",
         "struct u4 { uint32_t x, y, z, w; };", HOST_PRELUDE,
         'extern "C" void k_fuzz_host(const uint32_t *in, uint32_t *out, uint32_t *table) {',
         "    std::vector<uint32_t> a(N), b(N), c(N), d(N); std::vector<uint64_t> p(N), q(N);",
         "    for (uint32_t i = 0; i < N; ++i) {",
         "        a[i] = in[i]; b[i] = in[i + N]; c[i] = in[2 * N + (i ^ 1u)]; d[i] = in[3 * N + i];",
         "        const uint32_t *f = in + 4 * N + (i & ~3u);",
         "        p[i] = ((uint64_t) f[0] << 32) | f[3]; q[i] = ((uint64_t) f[1] << 32) | (f[2] ^ in[5 * N + i]);",
         "    }"]

    def local(s, ind):
        pad = "    " * ind
        k = s[0]
        if k in ("set", "upd"):
            o.append(f"{pad}{s[1]}[i] = {_fmt(s[3], True)};")
        elif k == "if":
            o.append(f"{pad}if ({_fmt(s[1], True)}) {{")
            for x in s[2]:
                local(x, ind + 1)
            if s[3]:
                o.append(f"{pad}}} else {{")
                for x in s[3]:
                    local(x, ind + 1)
            o.append(f"{pad}}}")
        elif k == "loop":
            _, acc, trip, body, stop = s
            o.append(f"{pad}for (uint32_t k = 0, n = {_fmt(trip, True)}; k < n; ++k) {{")
            o.append(f"{pad}    {acc}[i] = {_fmt(body, True)};")
            if stop:
                o.append(f"{pad}    if ({_fmt(stop, True)}) break;")
            o.append(f"{pad}}}")

    flat = []
    for s in g.stmts:
        flat += s[1] if s[0] == "multi" else [s]
    for s in flat:
        k = s[0]
        if k == "set":
            o.append(f"    std::vector<{'uint64_t' if s[2] else 'uint32_t'}> {s[1]}(N);")
        if k in ("set", "upd", "if", "loop"):
            o.append("    for (uint32_t i = 0; i < N; ++i) {")
            local(s, 2)
            o.append("    }")
        elif k == "cross":
            _, name, wide, src, how, delta, width, sel = s
            ty = "uint64_t" if wide else "uint32_t"
            o.append(f"    std::vector<{ty}> {name}(N);")
            o.append("    for (uint32_t w0 = 0; w0 < N; w0 += 64) for (uint32_t l = 0; l < 64; ++l) {")
            o.append(f"        const uint32_t W = {width}, seg = l & ~(W - 1u), r = l & (W - 1u); uint32_t from;")
            if how == "xor":
                o.append(f"        {{ const uint32_t j = l ^ {delta}u; from = j < seg + W ? j : l; }}")   # HIP: index = lane ^ mask; beyond the segment -> self
            elif how == "up":
                o.append(f"        from = r >= {delta}u ? l - {delta}u : l;")
            elif how == "down":
                o.append(f"        from = r + {delta}u < W ? l + {delta}u : l;")
            elif how == "idx":
                o.append(f"        from = seg + ({delta}u & (W - 1u));")
            elif how == "bperm":
                o.append(f"        from = {sel}[w0 + l] & 63u;")
            elif how == "perm":   # a push along a permutation of the lanes: lane s sends to (s * odd + 5) & 63, so l receives from its inverse
                o.append(f"        from = 0; for (uint32_t s2 = 0; s2 < 64; ++s2) if (((s2 * {2 * delta + 1}u + 5u) & 63u) == l) from = s2;")
            elif how == "readlane":
                o.append(f"        from = {delta}u;")
            elif how == "first":
                o.append("        from = 0;")
            else:
                o.append(f"        from = seg + ((({sel}[w0 + l] >> 3) & 63u) & (W - 1u));")
            o.append(f"        {name}[w0 + l] = {src}[w0 + from];")
            o.append("    }")
        elif k == "dpp":
            _, name, src, other, ctrl, rm, bm, bc, form = s
            o.append(f"    std::vector<uint32_t> {name}(N);")
            o.append("    for (uint32_t w0 = 0; w0 < N; w0 += 64) for (uint32_t l = 0; l < 64; ++l) { const uint32_t i = w0 + l;")
            o.append(f"        {name}[i] = {_dpp_text(s, True)}; }}")
        elif k == "ballot":
            o.append(f"    std::vector<uint64_t> {s[1]}(N);")
            o.append("    for (uint32_t w0 = 0; w0 < N; w0 += 64) { uint64_t m = 0;")
            o.append(f"        for (uint32_t l = 0; l < 64; ++l) m |= (uint64_t) (({s[2]}[w0 + l] >> {s[3]}u) & 1u) << l;")
            o.append(f"        for (uint32_t l = 0; l < 64; ++l) {s[1]}[w0 + l] = m; }}")
        elif k == "ifballot":
            o.append(f"    std::vector<uint64_t> {s[1]}(N, 0);")
            o.append("    for (uint32_t w0 = 0; w0 < N; w0 += 64) { uint64_t m = 0;")
            o.append(f"        for (uint32_t l = 0; l < 64; ++l) {{ const uint32_t i = w0 + l; if ({_fmt(s[2], True)}) m |= (uint64_t) (({s[3]}[i] >> {s[4]}u) & 1u) << l; }}")
            o.append(f"        for (uint32_t l = 0; l < 64; ++l) {{ const uint32_t i = w0 + l; if ({_fmt(s[2], True)}) {s[1]}[i] = m; }} }}")
        elif k == "ifvote":
            o.append(f"    std::vector<uint32_t> {s[1]}(N, 7u);")
            o.append("    for (uint32_t w0 = 0; w0 < N; w0 += 64) { uint32_t n = 0, in = 0;")
            o.append(f"        for (uint32_t l = 0; l < 64; ++l) {{ const uint32_t i = w0 + l; if ({_fmt(s[2], True)}) {{ ++in; n += {s[3]}[i] > {s[5]:#x}u; }} }}")
            o.append(f"        for (uint32_t l = 0; l < 64; ++l) {{ const uint32_t i = w0 + l; if ({_fmt(s[2], True)}) {s[1]}[i] = {'n != 0' if s[4] == 'any' else 'n == in'}; }} }}")
        elif k == "waveloop":
            o.append(f"    std::vector<uint32_t> {s[1]}(N, 0);")
            o.append(f"    for (uint32_t w0 = 0; w0 < N; w0 += 64) {{ uint32_t rest[64]; for (uint32_t l = 0; l < 64; ++l) rest[l] = {s[2]}[w0 + l] & {s[4]:#x}u;")
            o.append("        for (;;) { bool any = false; for (uint32_t l = 0; l < 64; ++l) any |= rest[l] != 0u; if (!any) break;")
            o.append(f"            for (uint32_t l = 0; l < 64; ++l) {{ {s[1]}[w0 + l] += (rest[l] & 1u) + 1u; rest[l] >>= {s[3]}u; }} }} }}")
        elif k == "vote":
            o.append(f"    std::vector<uint32_t> {s[1]}(N);")
            o.append("    for (uint32_t w0 = 0; w0 < N; w0 += 64) { uint32_t n = 0;")
            o.append(f"        for (uint32_t l = 0; l < 64; ++l) n += {s[2]}[w0 + l] > {s[4]:#x}u;")
            o.append(f"        for (uint32_t l = 0; l < 64; ++l) {s[1]}[w0 + l] = {'n != 0' if s[3] == 'any' else 'n == 64'}; }}")
        elif k in ("lds32", "lds64"):
            _, name, src, mw, aw, mr, ar = s
            ty = "uint32_t" if k == "lds32" else "uint64_t"
            o.append(f"    std::vector<{ty}> {name}(N);")
            o.append(f"    for (uint32_t g0 = 0; g0 < N; g0 += 256) {{ {ty} lds[256];")
            o.append(f"        for (uint32_t t = 0; t < 256; ++t) lds[(t * {mw}u + {aw}u) & 255u] = {src}[g0 + t];")
            o.append(f"        for (uint32_t t = 0; t < 256; ++t) {name}[g0 + t] = lds[(t * {mr}u + {ar}u) & 255u]; }}")
        elif k == "atomic":
            _, how, idx, val, mask = s
            opx = {"add": "+=", "or": "|=", "and": "&=", "xor": "^="}.get(how)
            o.append("    for (uint32_t i = 0; i < N; ++i) {")
            o.append(f"        uint32_t &slot = table[{ATOMIC_BASE[how]} + (({idx}[i] >> 5) & {mask}u)];")
            if opx:
                o.append(f"        slot {opx} {val}[i]; }}")
            else:
                o.append(f"        slot = {val}[i] {'>' if how == 'max' else '<'} slot ? {val}[i] : slot; }}")
        elif k == "ldspair":
            _, name, src, mw, aw, dist = s
            o.append(f"    std::vector<uint64_t> {name}(N);")
            o.append("    for (uint32_t g0 = 0; g0 < N; g0 += 256) { uint64_t lds[256];")
            o.append(f"        for (uint32_t t = 0; t < 256; ++t) lds[(t * {mw}u + {aw}u) & 255u] = {src}[g0 + t];")
            o.append(f"        for (uint32_t t = 0; t < 256; ++t) {name}[g0 + t] = lds[t & 127u] - lds[(t & 127u) + {dist}u]; }}")
        elif k == "lds128":
            _, name, srcs, mw, aw, mr, ar = s
            o.append(f"    std::vector<uint32_t> {name}(N);")
            o.append("    for (uint32_t g0 = 0; g0 < N; g0 += 256) { u4 lds[256];")
            o.append(f"        for (uint32_t t = 0; t < 256; ++t) lds[(t * {mw}u + {aw}u) & 255u] = u4{{{', '.join(x + '[g0 + t]' for x in srcs)}}};")
            o.append(f"        for (uint32_t t = 0; t < 256; ++t) {{ const u4 z = lds[(t * {mr}u + {ar}u) & 255u]; {name}[g0 + t] = z.x ^ (z.y << 1) ^ (z.z >> 1) ^ (z.w * 3u); }} }}")
    stride = len(g.v32) + 2 * len(g.v64) + 3
    o.append("    for (uint32_t i = 0; i < N; ++i) { uint32_t h = 0x811c9dc5u; uint64_t hh = 0xcbf29ce484222325ull;")
    o.append(f"        uint32_t *r = out + (uint64_t) i * {stride};")
    for idx, v in enumerate(g.v32):
        o.append(f"        h = (h ^ {v}[i]) * 0x01000193u; r[{idx}] = {v}[i];")
    base = len(g.v32)
    for idx, v in enumerate(g.v64):
        o.append(f"        hh = (hh ^ {v}[i]) * 0x100000001b3ull; r[{base + 2 * idx}] = (uint32_t) {v}[i]; r[{base + 2 * idx + 1}] = (uint32_t) ({v}[i] >> 32);")
    o.append(f"        r[{stride - 3}] = h; r[{stride - 2}] = (uint32_t) hh; r[{stride - 1}] = (uint32_t) (hh >> 32); }}")
    if g.chain and g.chain[2] == "packed":
        mix = g.chain[0]
        o.append(f"    for (uint32_t b = 0, sum = 0; b < {GRID}u; ++b) {{ sum += in[b * 7u + 3u] ^ {mix:#x}u; table[{PAIR_SLOT} + 2u * b] = sum; table[{PAIR_SLOT} + 2u * b + 1u] = 1u; }}")
    elif g.chain:
        mix = g.chain[0]
        o.append(f"    for (uint32_t b = 0, sum = 0; b < {GRID}u; ++b) {{ sum += in[b * 7u + 3u] ^ {mix:#x}u; table[{SUM_SLOT} + b] = sum; table[{FLAG_SLOT} + b] = 1u; }}")
    if g.ticket:
        mw, aw, mr, ar, mix, reads = g.ticket
        o.append(f"    for (uint32_t tile = 0; tile < {TILES}u; ++tile) {{ uint32_t lds[256];")
        o.append(f"        for (uint32_t t = 0; t < 256; ++t) lds[(t * {mw}u + {aw}u) & 255u] = in[(tile * 97u + t * 3u) % {NIN * N}u] ^ {mix:#x}u;")
        o.append(f"        for (uint32_t t = 0; t < 256; ++t) {{ uint32_t rv = lds[(t * {mr}u + {ar}u) & 255u];")
        if reads >= 2:
            o.append("            rv += lds[(t + 1u) & 255u] >> 3;")
        if reads >= 3:
            o.append("            rv ^= lds[(t ^ 32u) & 255u] * 5u;")
        o.append(f"            out[(uint64_t) N * {stride} + tile * 256u + t] = rv + tile; }} }}")
        o.append(f"    table[{TICKET_SLOT}] += {TILES}u + {GRID}u;   // every workgroup draws one number too many")
    o.append("}")
    return "\n".join(o) + "\n", stride


# The third leg (--model): the SAME device text, compiled for the host against tests/wavesim (the functional wave64 model the
# sanitizer and large-array rehearsals of the product run on).  What the model's header lacks of this tool's vocabulary is defined
# here out of its primitives (wave_exchange among the active lanes, wave_ballot).
MODEL_PRELUDE = r"""
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
struct uint4 { uint32_t x, y, z, w; };
static inline uint4 make_uint4(uint32_t x, uint32_t y, uint32_t z, uint32_t w) { return uint4{x, y, z, w}; }
template<typename T> static inline T __shfl_down(T v, unsigned d, int width = 64) {
    const int self = wavesim_lane();
    return wavesim_shfl_src(v, (self & (width - 1)) + static_cast<int>(d) >= width ? self : self + static_cast<int>(d));
}
static inline uint32_t atomicMax(uint32_t *p, uint32_t v) { return __atomic_fetch_max(p, v, __ATOMIC_SEQ_CST); }
static inline uint32_t atomicMin(uint32_t *p, uint32_t v) { return __atomic_fetch_min(p, v, __ATOMIC_SEQ_CST); }
static inline uint32_t atomicAnd(uint32_t *p, uint32_t v) { return __atomic_fetch_and(p, v, __ATOMIC_SEQ_CST); }
static inline uint32_t atomicXor(uint32_t *p, uint32_t v) { return __atomic_fetch_xor(p, v, __ATOMIC_SEQ_CST); }
static inline int __any(int p) { return __ballot(p != 0) != 0; }
static inline int __all(int p) { return __ballot(p == 0) == 0; }
#define __lane_u32 static_cast<uint32_t>(wavesim_lane())
#define __builtin_amdgcn_alignbyte(hi, lo, s) static_cast<uint32_t>(((static_cast<uint64_t>(hi) << 32) | static_cast<uint64_t>(lo)) >> (8u * ((s) & 3u)))
static inline uint32_t __builtin_amdgcn_ubfe(uint32_t v, uint32_t off, uint32_t n) { off &= 31u; n &= 31u; return n ? (v >> off) & ((1u << n) - 1u) : 0u; }
static inline int32_t __builtin_amdgcn_sbfe(int32_t v, uint32_t off, uint32_t n) { off &= 31u; n &= 31u; if (!n) return 0;
    const uint32_t f = static_cast<uint32_t>(v >> off) & ((1u << n) - 1u); return (f >> (n - 1u)) & 1u ? static_cast<int32_t>(f | ~((1u << n) - 1u)) : static_cast<int32_t>(f); }
#define __builtin_amdgcn_mbcnt_lo(m, add) (static_cast<uint32_t>(__builtin_popcount((m) & (__lane_u32 >= 32u ? 0xffffffffu : (1u << __lane_u32) - 1u))) + (add))
#define __builtin_amdgcn_mbcnt_hi(m, add) (static_cast<uint32_t>(__builtin_popcount((m) & (__lane_u32 <= 32u ? 0u : (1u << (__lane_u32 - 32u)) - 1u))) + (add))
#define __builtin_amdgcn_ds_bpermute(addr, v) static_cast<int>(wavesim_shfl_src(static_cast<uint32_t>(v), (static_cast<uint32_t>(addr) >> 2) & 63))
#define __builtin_amdgcn_readfirstlane(v) static_cast<int>(wavesim_shfl_src(static_cast<uint32_t>(v), __builtin_ctzll(__ballot(true))))
"""

MODEL_MAIN = r"""
int main(int argc, char **argv) {
    if (argc != 6) return 2;
    const size_t nin = strtoull(argv[3], nullptr, 0), nout = strtoull(argv[4], nullptr, 0), zero_from = strtoull(argv[5], nullptr, 0);
    std::vector<uint32_t> in(nin), out(nout, 0xDEADBEEFu), table(TABLE_WORDS_, 0u);
    for (int k = 0; k < 16; ++k) table[MIN_BASE_ + k] = table[AND_BASE_ + k] = 0xffffffffu;
    for (size_t k = zero_from; k < nout; ++k) out[k] = 0;
    FILE *f = fopen(argv[1], "rb"); if (!f || fread(in.data(), 4, nin, f) != nin) return 3; fclose(f);
    hipLaunchKernelGGL(k_fuzz, dim3(GRID_), dim3(256), 0, nullptr, static_cast<const uint32_t *>(in.data()), out.data(), table.data());
    f = fopen(argv[2], "wb"); if (!f || fwrite(out.data(), 4, nout, f) != nout || fwrite(table.data(), 4, table.size(), f) != table.size()) return 4; fclose(f);
    return 0;
}
"""


def model_source(g) -> str:
    dev = device_source(g).replace("#include <hip/hip_runtime.h>\n#include <cstdint>\n", "").replace('extern "C" __global__', "__global__")
    main = MODEL_MAIN.replace("GRID_", str(GRID)).replace("TABLE_WORDS_", str(TABLE_WORDS)).replace("MIN_BASE_", str(ATOMIC_BASE["min"])).replace("AND_BASE_", str(ATOMIC_BASE["and"]))
    return MODEL_PRELUDE + dev + main


_MODEL_OBJ = {}


def run_model(g, seed: int, workdir: str, x: np.ndarray, stride: int):
    """(status, output words): the case on tests/wavesim, under a seeded random fiber schedule"""
    wdir = os.path.join(ROOT, "tests", "wavesim")
    obj = _MODEL_OBJ.get(workdir)
    if obj is None:
        obj = os.path.join(workdir, "wavesim_model.o")
        r = subprocess.run([CLANG, "-std=c++17", "-O1", "-pthread", "-Wno-unknown-attributes", "-I", wdir, "-c", os.path.join(wdir, "wavesim.cc"), "-o", obj], capture_output=True, text=True)
        if r.returncode:
            return "model-compile", r.stderr[-1500:]
        _MODEL_OBJ[workdir] = obj
    src, exe = os.path.join(workdir, f"case{seed}_model.cc"), os.path.join(workdir, f"case{seed}_model")
    open(src, "w").write(model_source(g))
    r = subprocess.run([CLANG, "-std=c++17", "-O1", "-pthread", "-Wno-unknown-attributes", "-Wno-shift-count-overflow", "-I", wdir, src, obj, "-ldl", "-o", exe], capture_output=True, text=True)
    if r.returncode:
        return "model-compile", r.stderr[-1500:]
    fin, fout = os.path.join(workdir, f"case{seed}.in"), os.path.join(workdir, f"case{seed}.out")
    x.tofile(fin)
    r = subprocess.run([exe, fin, fout, str(x.size), str(N * stride + TILES * 256), str(N * stride + TILES * 256 if g.ticket else N * stride)], capture_output=True, text=True, timeout=300, env={**os.environ, "WAVESIM_SCHEDULE": f"random:{seed % 1000}"})
    if r.returncode:
        return "model-run", f"rc {r.returncode}: {r.stderr[-800:]}"
    return "ok", np.fromfile(fout, dtype=np.uint32)


class Hardware:
    """The same code object on a real gfx950 device (tests/test_zz_hip_fuzz_hardware.py, tools/gpu_batch.sh explain): loaded with the HIP
    module API through ctypes, buffers from torch.  Raises RuntimeError on any plumbing failure -- the callers tell that apart from a
    result that differs."""

    def __init__(self):
        import torch
        if not torch.cuda.is_available():
            raise RuntimeError("no GPU")
        self.torch = torch
        torch.zeros(1, device="cuda")  # (the runtime and its primary context)
        # the SAME runtime library torch has loaded (it ships its own copy: a second one in the process would not know torch's pointers)
        path = "libamdhip64.so"
        with open("/proc/self/maps") as f:
            for line in f:
                if "libamdhip64" in line:
                    path = line.split()[-1]
                    break
        self.hip = C.CDLL(path)
        self.hip.hipGetErrorString.restype = C.c_char_p
        self.hip.hipModuleLoad.argtypes = [C.c_void_p, C.c_char_p]
        self.hip.hipModuleGetFunction.argtypes = [C.c_void_p, C.c_void_p, C.c_char_p]
        self.hip.hipModuleLaunchKernel.argtypes = [C.c_void_p] + [C.c_uint] * 7 + [C.c_void_p, C.c_void_p, C.c_void_p]
        self.hip.hipModuleUnload.argtypes = [C.c_void_p]

    def _check(self, rc, what):
        if rc != 0:
            raise RuntimeError(f"{what}: HIP error {rc}: {self.hip.hipGetErrorString(rc).decode(errors='replace')}")

    def run(self, co_path: str, x: np.ndarray, nout: int, zero_from: int):
        """(output words incl. the tile region, table) after one launch of k_fuzz"""
        torch, hip = self.torch, self.hip
        d_in = torch.from_numpy(x.view(np.int32)).cuda()
        out0 = np.full(nout, 0xDEADBEEF, dtype=np.uint32)
        out0[zero_from:] = 0
        d_out = torch.from_numpy(out0.view(np.int32)).cuda()
        d_table = torch.from_numpy(fresh_table().view(np.int32)).cuda()
        mod, fn = C.c_void_p(), C.c_void_p()
        self._check(hip.hipModuleLoad(C.byref(mod), co_path.encode()), "hipModuleLoad")
        try:
            self._check(hip.hipModuleGetFunction(C.byref(fn), mod, b"k_fuzz"), "hipModuleGetFunction")
            args = C.create_string_buffer(struct.pack("<QQQ", d_in.data_ptr(), d_out.data_ptr(), d_table.data_ptr()))
            size = C.c_size_t(24)
            extra = (C.c_void_p * 5)(1, C.cast(args, C.c_void_p), 2, C.cast(C.pointer(size), C.c_void_p), 3)  # HIP_LAUNCH_PARAM_BUFFER_POINTER / _SIZE / _END
            torch.cuda.synchronize()
            self._check(hip.hipModuleLaunchKernel(fn, GRID, 1, 1, BLOCK, 1, 1, 0, None, None, extra), "hipModuleLaunchKernel")
            self._check(hip.hipDeviceSynchronize(), "hipDeviceSynchronize")
        finally:
            hip.hipModuleUnload(mod)
        return d_out.cpu().numpy().view(np.uint32), d_table.cpu().numpy().view(np.uint32)


def fresh_table() -> np.ndarray:
    t = np.zeros(TABLE_WORDS, dtype=np.uint32)
    t[ATOMIC_BASE["min"]:ATOMIC_BASE["min"] + 16] = 0xFFFFFFFF
    t[ATOMIC_BASE["and"]:ATOMIC_BASE["and"] + 16] = 0xFFFFFFFF
    return t


def inputs(rng: np.random.Generator) -> np.ndarray:
    x = rng.integers(0, 1 << 32, size=NIN * N, dtype=np.uint64).astype(np.uint32)
    # corner values sprinkled in: zeros, all-ones, sign bits, small numbers
    special = np.array([0, 1, 2, 0x7fffffff, 0x80000000, 0xffffffff, 0xfffffffe, 31, 32, 63, 64, 0x10000, 0xffff], dtype=np.uint32)
    where = rng.random(x.size) < 0.15
    x[where] = special[rng.integers(0, special.size, size=int(where.sum()))]
    return x


def run_case(seed: int, workdir: str, opt: str, nstmts: int, gx, intrinsics: bool = False, model: bool = False, hardware=None):
    """hardware: a Hardware() -- the code object also runs on the device; its words must be the INTERPRETER's (that is what the
    interpreter claims to be), whatever the host says: "HARDWARE-MISMATCH" otherwise"""
    g = Gen(random.Random(seed), intrinsics, model).build(nstmts)
    dev = device_source(g)
    host, stride = host_source(g)
    dpath, hpath = os.path.join(workdir, f"case{seed}.hip"), os.path.join(workdir, f"case{seed}_host.cc")
    open(dpath, "w").write(dev)
    open(hpath, "w").write(host)
    co, so = os.path.join(workdir, f"case{seed}.hsaco"), os.path.join(workdir, f"case{seed}_host.so")
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", f"-{opt}", "--genco", "--no-gpu-bundle-output", dpath, "-o", co], capture_output=True, text=True)
    if r.returncode:
        return "device-compile", r.stderr[-1500:]
    r = subprocess.run([CLANG, "-O1", "-shared", "-fPIC", "-std=c++17", hpath, "-o", so], capture_output=True, text=True)
    if r.returncode:
        return "host-compile", r.stderr[-1500:]
    x = inputs(np.random.default_rng(seed))
    want = np.zeros(N * stride + TILES * 256, dtype=np.uint32)
    L = C.CDLL(so)
    want_t = fresh_table()
    L.k_fuzz_host(C.c_void_p(x.ctypes.data), C.c_void_p(want.ctypes.data), C.c_void_p(want_t.ctypes.data))
    want = np.concatenate([want, want_t])
    if model:
        st, mgot = run_model(g, seed, workdir, x, stride)
        if st != "ok":
            return st, mgot
        if not np.array_equal(mgot, want):
            bad = np.flatnonzero(mgot != want)
            names = g.v32 + [f"{v}.{h}" for v in g.v64 for h in ("lo", "hi")] + ["h", "hh.lo", "hh.hi"]
            first = int(bad[0])
            where = f"work-item {first // stride} {names[first % stride]}" if first < N * stride else f"tile word {first - N * stride}" if first < N * stride + TILES * 256 else f"table[{first - N * stride - TILES * 256}]"
            return "MODEL-MISMATCH", f"{bad.size} words differ; first: {where} model {mgot[first]:#x} want {want[first]:#x}"
    k = gx.Kernel(gx.CodeObject(co), "k_fuzz")
    if k.missing:
        return "unknown-op", sorted(k.missing)
    got, got_t = np.full(N * stride + TILES * 256, 0xDEADBEEF, dtype=np.uint32), fresh_table()
    if not g.ticket:
        got[N * stride:] = 0
    lds = k.lds_bytes if hasattr(k, "lds_bytes") else 0
    try:
        # (the schedule varies with the case: index order, reverse, reshuffled every pass -- a race shows only when its loser runs first)
        gx.run_grid(k, GRID, BLOCK, lds, struct.pack("<QQQ", x.ctypes.data, got.ctypes.data, got_t.ctypes.data), resident=2, quantum=(400, 60, 2500)[seed % 3],
                    order=("forward", "reverse", f"random:{seed}")[(seed // 3) % 3])
        got = np.concatenate([got, got_t])
    except gx.Unsupported as e:
        return "unsupported", str(e)
    except Exception as e:  # a hazard / wait report on compiler output, or an interpreter fault
        return "interpreter-error", f"{type(e).__name__}: {e}"
    if hardware is not None:
        h_out, h_table = hardware.run(co, x, N * stride + TILES * 256, N * stride + TILES * 256 if g.ticket else N * stride)
        hgot = np.concatenate([h_out, h_table])
        if not np.array_equal(hgot, got):
            import audit_machine_sink
            if any(b for _, _, _, b in audit_machine_sink.audit(dpath, [f"-{opt}"], workdir)):
                return "compiler-sunk-load", "a load moved across a barrier: device and interpreter may each run the race their own way"
            bad = np.flatnonzero(hgot != got)
            return "HARDWARE-MISMATCH", f"{bad.size} words differ between the device and the interpreter; first at {int(bad[0])}: device {hgot[bad[0]]:#x} interpreter {got[bad[0]]:#x} host {want[bad[0]]:#x}"
    if np.array_equal(got, want):
        return "ok", {x.op for x in k.code.values()}
    # a wrong answer that is the COMPILER's: this image's LLVM can sink an LDS load past __syncthreads() (tools/audit_machine_sink.py)
    import audit_machine_sink
    sunk = [b for _, _, _, b in audit_machine_sink.audit(dpath, [f"-{opt}"], workdir) if b]
    if sunk:
        return "compiler-sunk-load", f"{len(sunk)} load(s) moved across a barrier by machine-sink; the interpreter ran the race to a different answer"
    # ... or a disagreement between the compiler's two instruction selectors: the same source through GlobalISel.  If that build
    # runs to the host's answer, SelectionDAG's code computes something else from the same IR (seen: `or i64 x, ~zext(i32)` with a
    # second use of the ~zext gets the wrong high half; a byte gather over `ashr (perm builtin), 8k` reads low bytes for sign bytes --
    # docs/compiler_findings.md, findings 2 and 4; tests/test_compiler_sink_audit.py holds both reproducers)
    co2 = os.path.join(workdir, f"case{seed}_gisel.hsaco")
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", f"-{opt}", "--genco", "--no-gpu-bundle-output", "-mllvm", "-global-isel", "-mllvm", "-global-isel-abort=2",
                        dpath, "-o", co2], capture_output=True, text=True)
    if r.returncode == 0:
        k2 = gx.Kernel(gx.CodeObject(co2), "k_fuzz")
        if not k2.missing:
            got2, got2_t = np.full(N * stride + TILES * 256, 0xDEADBEEF, dtype=np.uint32), fresh_table()
            if not g.ticket:
                got2[N * stride:] = 0
            try:
                gx.run_grid(k2, GRID, BLOCK, 0, struct.pack("<QQQ", x.ctypes.data, got2.ctypes.data, got2_t.ctypes.data), resident=2, quantum=400)
                if np.array_equal(np.concatenate([got2, got2_t]), want):
                    return "codegen-disagreement", "the GlobalISel build of the same source runs to the host's answer, the SelectionDAG build does not"
            except Exception:
                pass
    # ... or the compiler's v_bitop3 formation (tools/audit_bitop3.py): both selectors share the matcher that gets the truth table wrong when
    # an inner bitwise value is reached twice from one root -- if this case's IR has that shape, the wrong answer is put down to it
    import audit_bitop3
    hits, _ = audit_bitop3.audit(dpath, [f"-{opt}"], workdir)
    if hits and audit_bitop3.emitted_table(audit_bitop3.REPRODUCER_IR, workdir) != audit_bitop3.REPRODUCER_TABLE:
        return "compiler-bitop3", f"{len(hits)} bitwise expression(s) of the shape this compiler fuses into a wrong v_bitop3 table (e.g. {hits[0][1]} reaches {hits[0][2]} twice)"
    bad = np.flatnonzero(got != want)
    cols = sorted({int(b % stride) for b in bad})
    names = g.v32 + [f"{v}.{h}" for v in g.v64 for h in ("lo", "hi")] + ["h", "hh.lo", "hh.hi"]
    first = int(bad[0])
    where = f"work-item {first // stride} {names[first % stride]}" if first < N * stride else f"tile word {first - N * stride}" if first < N * stride + TILES * 256 else f"table[{first - N * stride - TILES * 256}]"
    return "MISMATCH", f"{bad.size} words differ; columns {[names[c] for c in cols][:8]}; first: {where} got {got[first]:#x} want {want[first]:#x}"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--cases", type=int, default=20)
    ap.add_argument("--opt", default="O3", choices=["O1", "O2", "O3", "Os"])
    ap.add_argument("--statements", type=int, default=28)
    ap.add_argument("--intrinsics", action="store_true", help="also draw from the gfx950 builtins the product's kernels use (host meaning: HOST_PRELUDE)")
    ap.add_argument("--model", action="store_true", help="a third leg: the same device text on the functional model (tests/wavesim) under a random fiber schedule")
    ap.add_argument("--hardware", action="store_true", help="a fourth leg on a GPU box: the code object on the device; its words must be the interpreter's")
    ap.add_argument("--ops-out", default=None, help="append the opcodes executed in agreeing kernels to this file (one per line)")
    ap.add_argument("--keep", default=None, help="directory for the generated sources (default: a temporary one)")
    args = ap.parse_args()
    from tests import gfx950_exec as gx

    hardware = Hardware() if args.hardware else None
    work = args.keep or tempfile.mkdtemp(prefix="fuzz_ivc_")
    os.makedirs(work, exist_ok=True)
    tally, ops_seen, unknown, problems = collections.Counter(), set(), collections.Counter(), []
    t0 = time.time()
    for n in range(args.cases):
        seed = args.seed * 100000 + n
        status, info = run_case(seed, work, args.opt, args.statements, gx, args.intrinsics, args.model, hardware)
        tally[status] += 1
        if status == "ok":
            ops_seen |= info
            if not args.keep:
                for f in os.listdir(work):
                    if f.startswith(f"case{seed}"):
                        os.unlink(os.path.join(work, f))
        elif status == "unknown-op":
            unknown.update(info)
        else:
            problems.append((seed, status, info))
            print(f"case {seed}: {status}: {info}", flush=True)
    print(f"seed {args.seed} -{args.opt}{' +intrinsics' if args.intrinsics else ''}{' +model' if args.model else ''}{' +hardware' if args.hardware else ''}: {dict(tally)} in {time.time() - t0:.0f} s; {len(ops_seen)} distinct opcodes executed in agreeing kernels")
    if args.ops_out:
        with open(args.ops_out, "a") as f:
            f.write("".join(o + "\n" for o in sorted(ops_seen)))
    if unknown:
        print("  opcodes the interpreter does not know (kernels skipped):", dict(unknown.most_common()))
    if problems:
        print(f"  sources of the {len(problems)} problem cases kept in {work}")
    return 1 if any(s in ("MISMATCH", "MODEL-MISMATCH", "HARDWARE-MISMATCH", "interpreter-error", "model-run", "model-compile") for _, s, _ in problems) else 0


if __name__ == "__main__":
    sys.exit(main())
