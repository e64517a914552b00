"""The wait states of the kernels' inline assembly, checked by LLVM's own gfx950 hazard recogniser (tools/asm_hazards.py; CPU only).

hipcc does not look inside an `asm volatile` statement.  Rounds 3-4 counted the s_nops of ndzip_amd/csrc/gfx950_lds.hpp by hand against
the builder's reading of the ISA manual, and the interpreter that executed them (tests/gfx950_exec.py) checked them against the same
reading -- which missed a gfx940-family rule (VALU writes SGPR / VCC -> VALU reads it: 2 wait states): the three f64 DPP helpers of
round 4 would have consumed stale carries / select masks on silicon.  Here every asm block of the BUILT code is translated to MIR and
run through `llc -run-pass post-RA-hazard-rec`; LLVM inserting a single S_NOP fails the test.  Also: the new 64-bit EXEC-masked plane
compaction, as hipcc assembled it, executed by the interpreter under random entry EXEC masks against a plain Python model."""
import os
import re
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import asm_hazards  # noqa: E402

pytestmark = pytest.mark.skipif(not os.path.exists(asm_hazards.LLC), reason="llc of the ROCm toolchain")


@pytest.fixture(scope="module")
def listings():
    return asm_hazards.listings_of_build()


def test_llvm_finds_no_hazard_in_the_built_inline_assembly(listings):
    blocks = sum(len(asm_hazards.asm_blocks(l)) for l in listings.values())
    assert blocks > 100, blocks  # (both translation units are full of them: compaction, DPP scans, pins)
    assert asm_hazards.check(listings) == []
    text = "\n".join("\n".join(l) for l in listings.values())
    for opcode in ("v_cmpx_ne_u32_e32", "v_cmpx_gt_i32_e32", "v_add_co_u32_dpp", "v_addc_co_u32_dpp", "v_cndmask_b32_dpp"):
        assert opcode in text, opcode  # (the blocks this test is about are really in the build)


def _listing(*blocks):
    out = ["_Z4testv:"]
    for b in blocks:
        out += ["\t;;#ASMSTART"] + ["\t" + l for l in b] + ["\t;;#ASMEND"]
    return {"synthetic": out}


def test_llvm_flags_the_sequences_round_4_shipped():
    dpp = "row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"
    carry_back_to_back = ["s_nop 1", f"v_add_co_u32_dpp v1, vcc, v1, v1 {dpp}", f"v_addc_co_u32_dpp v2, vcc, v2, v2, vcc {dpp}", "s_nop 1"]
    carry_padded = ["s_nop 1", f"v_add_co_u32_dpp v1, vcc, v1, v1 {dpp}", "s_nop 1", f"v_addc_co_u32_dpp v2, vcc, v2, v2, vcc {dpp}", "s_nop 1"]
    select_back_to_back = ["s_nop 1", "v_cmp_ne_u32_e32 vcc, 0, v9", "v_cndmask_b32_dpp v1, v2, v3, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf", "s_nop 1"]
    select_padded = select_back_to_back[:2] + ["s_nop 1"] + select_back_to_back[2:]
    assert asm_hazards.check(_listing(carry_back_to_back)) != []
    assert asm_hazards.check(_listing(carry_padded)) == []
    assert asm_hazards.check(_listing(select_back_to_back)) != []
    assert asm_hazards.check(_listing(select_padded)) == []
    # boundaries: a DPP read of compiled code's last VALU result in front / of the block's last result behind
    assert asm_hazards.check(_listing(carry_padded[1:])) != []    # no s_nop in front
    assert asm_hazards.check(_listing(carry_padded[:-1])) != []   # no s_nop behind
    # v_cmpx (VALU write of EXEC) -> a DPP instruction of the compiled code behind the block: 5 wait states
    tail = ["ds_write_b32 v9, v8", "v_add_u32_e32 v9, 4, v9", "s_mov_b64 exec, s[4:5]"]
    assert asm_hazards.check(_listing(["v_cmpx_ne_u32_e32 vcc, 0, v8"] + tail)) != []
    assert asm_hazards.check(_listing(["v_cmpx_ne_u32_e32 vcc, 0, v8"] + tail + ["s_nop 1"])) == []
    # an opcode the translator does not know is an error, not a pass
    assert any("no MIR translation" in p for p in asm_hazards.check(_listing(["v_permlane32_swap_b32 v1, v2"])))


def test_a_valu_written_scalar_operand_is_reported():
    listing = {"synthetic": ["_Z4testv:", "\tv_readfirstlane_b32 s6, v0", "\t;;#ASMSTART", "\tv_bitop3_b32 v1, v1, v2, s6 bitop3:0x6c", "\ts_nop 1", "\t;;#ASMEND"]}
    assert any("written by a VALU instruction" in p for p in asm_hazards.check(listing))
    listing["synthetic"][1] = "\ts_movk_i32 s6, 0x70"
    assert asm_hazards.check(listing) == []


def _model_append64(a, flags, w):
    """lds_append_flagged64 for one lane: [(LDS byte address, word)], final a"""
    stores = []
    for i in range(32):
        if (flags >> (31 - i)) & 1:
            stores.append((a ^ ((a >> 3) & 0x70), int(w[i])))
            a += 8
    return stores, a


@pytest.mark.parametrize("seed", range(6))
def test_f64_append_block_as_assembled_under_random_entry_exec(listings, seed):
    """the four asm statements of one lds_append_flagged64 call, with the registers hipcc gave them, interpreted with a random set
    of active lanes: active lanes store exactly the flagged words at the swizzled addresses, inactive lanes store nothing and keep
    their registers, EXEC is back at the entry mask behind every statement"""
    from tests import gfx950_exec as gx

    blocks = [b for b in asm_hazards.asm_blocks(listings["kernels_f64.hip"]) if any(l.startswith("v_cmpx_gt_i32_e32") for l in b[1])]
    assert len(blocks) >= 4 and len(blocks) % 4 == 0
    call = blocks[4 * (seed % (len(blocks) // 4)): 4 * (seed % (len(blocks) // 4)) + 4]
    rng = np.random.default_rng(seed)
    first = call[0][1]
    a_reg = int(re.match(r"v_lshrrev_b32_e32 v\d+, 3, v(\d+)", first[1]).group(1))
    f_reg = int(re.match(r"v_cmpx_gt_i32_e32 vcc, 0, v(\d+)", first[0]).group(1))
    full = re.match(r"s_mov_b64 exec, s\[(\d+):(\d+)\]", first[5])
    m_reg = int(re.match(r"v_bitop3_b32 v\d+, v\d+, v\d+, s(\d+) bitop3:0x6c", first[2]).group(1))
    wg = gx.Workgroup(0, 1, 65536)
    w = gx.Wave(wg, 0, {}, 0)
    entry = int(rng.integers(0, 1 << 63, dtype=np.uint64)) | (int(rng.integers(0, 2)) << 63) if seed else gx.M64
    w.exec = entry
    w.s[int(full.group(1))], w.s[int(full.group(2))] = entry & gx.M32, entry >> 32
    w.s[m_reg] = 0x70
    # every lane its own region of 512 bytes (two lanes of the product never share an address either)
    a0 = (np.arange(64, dtype=np.uint32) * 512 + rng.integers(0, 32, 64).astype(np.uint32) * 8 + (4 * rng.integers(0, 2, 64)).astype(np.uint32))
    flags = rng.integers(0, 1 << 32, 64, dtype=np.uint64).astype(np.uint32)
    flags[rng.integers(0, 64, 6)] = [0, 0xFFFFFFFF, 0x80000000, 1, 0xFFFF0000, 0x0000FFFF]
    w.v[a_reg], w.v[f_reg] = a0, flags
    words = np.zeros((32, 64), dtype=np.uint32)
    lds_before = wg.lds.copy()
    for k, (_, lines, _) in enumerate(call):
        regs = [int(re.match(r"ds_write_b32 v\d+, v(\d+)", l).group(1)) for l in lines if l.startswith("ds_write_b32")]
        assert len(regs) == 8
        for j, r in enumerate(regs):  # (a plane register may be reused by a later statement: load its words just in time)
            words[8 * k + j] = rng.integers(0, 1 << 32, 64, dtype=np.uint64).astype(np.uint32)
            w.v[r] = words[8 * k + j]
        for n, text in enumerate(lines):
            op, _, rest = text.partition(" ")
            args, mods = gx._split_operands(rest)
            ins = gx.Ins(op, args, mods, 4 * n, 4, text)
            ins.fn = gx.OPS[op]
            ins.fn(w, ins)
        assert w.exec == entry, k
    want = lds_before.copy()
    for lane in range(64):
        if (entry >> lane) & 1:
            stores, a_end = _model_append64(int(a0[lane]), int(flags[lane]), words[:, lane])
            for addr, word in stores:
                want[addr:addr + 4] = np.frombuffer(np.uint32(word).tobytes(), dtype=np.uint8)
            assert int(w.v[a_reg][lane]) == a_end, lane
        else:
            assert int(w.v[a_reg][lane]) == int(a0[lane]) and int(w.v[f_reg][lane]) == int(flags[lane]), lane
    assert np.array_equal(wg.lds, want)
