"""An independent reading of the DPP wave scan (no GPU, no functional model): LLVM's own AMDGPU atomic optimizer builds a wave64
inclusive scan for gfx9-family targets when it pre-reduces a divergent atomicAdd (strategy DPP) -- four row shifts, row_bcast:15
into rows 1 and 3, row_bcast:31 into rows 2 and 3.  The product's wave_inclusive_scan (ndzip_amd/csrc/codec_kernels.hpp) must
compile to the very same sequence of DPP controls, row masks, bank masks and bound_ctrl on gfx950: then its lane semantics are the
compiler writers' reading of the ISA, not only this repo's (tests/wavesim models them from the ISA manual; the GPU suite checks
them on silicon in test_hip_stages.py::test_wave_scan_and_sum)."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"

REFERENCE = r"""
#include <hip/hip_runtime.h>
__global__ void llvm_scan(int *p, const int *in, int *out) { out[threadIdx.x] = atomicAdd(p, in[threadIdx.x]); }
"""

PRODUCT = r"""
#include "codec_kernels.hpp"
__global__ void product_scan(const uint32_t *in, uint32_t *out) {
    out[threadIdx.x] = ndzip_hip::wave_inclusive_scan(in[threadIdx.x], static_cast<int>(threadIdx.x & 63u));
}
"""


def _dpp_sequence(asm, kernel):
    body = re.search(rf"^{kernel}:.*?s_endpgm", asm, re.S | re.M)
    if body is None:  # (mangled name)
        body = re.search(rf"^_Z\w*{kernel}\w*:.*?s_endpgm", asm, re.S | re.M)
    assert body, kernel
    seq = []
    for line in body.group(0).splitlines():
        m = re.search(r"v_add_u32_dpp\s+\S+,\s*\S+,\s*\S+\s+(.*)$", line.split(";")[0].rstrip())
        if m:
            seq.append(" ".join(m.group(1).split()))
    return seq


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc")
def test_wave_scan_is_the_sequence_llvm_generates_for_its_own_wave64_scan(tmp_path):
    out = {}
    for name, src, flags in (("llvm", REFERENCE, ["-mllvm", "-amdgpu-atomic-optimizer-strategy=DPP"]),
                             ("product", PRODUCT, ["-I", os.path.join(ROOT, "ndzip_amd", "csrc")])):
        f = tmp_path / f"{name}.hip"
        f.write_text(src)
        s = tmp_path / f"{name}.s"
        r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", *flags, str(f), "-o", str(s)],
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        out[name] = _dpp_sequence(s.read_text(), "llvm_scan" if name == "llvm" else "product_scan")
    want = ["row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1", "row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1",
            "row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1", "row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1",
            "row_bcast:15 row_mask:0xa bank_mask:0xf", "row_bcast:31 row_mask:0xc bank_mask:0xf"]
    assert out["llvm"] == want, out["llvm"]        # what LLVM emits for its own scan on gfx950 (wave64)
    assert out["product"] == out["llvm"], out["product"]


# ---- the five instruction-level assembly blocks of gfx950_lds.hpp against hipcc's own lowering of their NDZIP_NO_EXEC_ASM fallbacks ----
#
# Every block has a fallback in plain C++ / __builtin_amdgcn_update_dpp that means the same (ndzip_amd/_variants/plain.so is built from
# them).  Compiled for gfx950, the fallback is LLVM's reading of the operation: which DPP control, row mask, bank mask and bound_ctrl a
# step takes, which operand of a select the swapped lane goes to, which register an EXEC-masked store writes and to what address
# expression.  The hand assembly must say the same, control word by control word -- then its lane semantics are the compiler writers'
# reading of the ISA and not only this repository's (rounds 3-4: a misreading shared by the kernels, the functional model and the
# instruction-level interpreter passed every builder-authored check).  A one-token change of a control word in gfx950_lds.hpp fails here
# (test_a_changed_control_word_is_caught does exactly that to a copy).
HELPERS = r"""
#include "gfx950_lds.hpp"
using namespace ndzip_hip;
template<int D> __device__ void row_kernel(uint32_t *p) {
    uint32_t lo[8], hi[8];
    for (int j = 0; j < 8; ++j) { lo[j] = p[threadIdx.x * 16 + j]; hi[j] = p[threadIdx.x * 16 + 8 + j]; }
    row_scan_step64<D>(lo, hi);
    for (int j = 0; j < 8; ++j) { p[threadIdx.x * 16 + j] = lo[j]; p[threadIdx.x * 16 + 8 + j] = hi[j]; }
}
extern "C" __global__ void k_row1(uint32_t *p) { row_kernel<1>(p); }
extern "C" __global__ void k_row2(uint32_t *p) { row_kernel<2>(p); }
extern "C" __global__ void k_row4(uint32_t *p) { row_kernel<4>(p); }
extern "C" __global__ void k_row8(uint32_t *p) { row_kernel<8>(p); }
extern "C" __global__ void k_scan64(uint32_t *p) {
    uint32_t lo = p[threadIdx.x * 2], hi = p[threadIdx.x * 2 + 1];
    wave_inclusive_scan64(lo, hi);
    p[threadIdx.x * 2] = lo; p[threadIdx.x * 2 + 1] = hi;
}
extern "C" __global__ void k_pair(const uint32_t *in, uint32_t *out) {
    uint32_t a[4], b[4], lo[4], hi[4];
    for (int j = 0; j < 4; ++j) { a[j] = in[threadIdx.x * 8 + j]; b[j] = in[threadIdx.x * 8 + 4 + j]; }
    pair_exchange_select4(threadIdx.x & 1u, a, b, lo, hi);
    for (int j = 0; j < 4; ++j) { out[threadIdx.x * 8 + j] = lo[j]; out[threadIdx.x * 8 + 4 + j] = hi[j]; }
}
extern "C" __global__ void k_append(const uint32_t *p, uint32_t *out) {
    extern __shared__ char lds[];
    uint32_t w[32];
    for (int j = 0; j < 32; ++j) w[j] = p[threadIdx.x * 32 + j];
    uint32_t a = lds_append_nonzero(lds_address(lds) + threadIdx.x * 128, w);
    lds_append_complete();
    __syncthreads();
    out[threadIdx.x] = a + *reinterpret_cast<uint32_t *>(lds + threadIdx.x * 4);
}
extern "C" __global__ void k_append64(const uint32_t *p, uint32_t *out) {
    extern __shared__ char lds[];
    uint32_t w[32];
    for (int j = 0; j < 32; ++j) w[j] = p[threadIdx.x * 36 + j];
    lds_append_flagged64(lds_address(lds) + threadIdx.x * 256, p[threadIdx.x * 36 + 32], w);
    lds_append_complete();
    __syncthreads();
    out[threadIdx.x] = *reinterpret_cast<uint32_t *>(lds + threadIdx.x * 4);
}
"""

CSRC = os.path.join(ROOT, "ndzip_amd", "csrc")


def _compile(tmp_path, name, include_dir, defines=()):
    f = tmp_path / f"{name}.hip"
    f.write_text(HELPERS)
    s = tmp_path / f"{name}.s"
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", *[f"-D{d}" for d in defines], "-I", include_dir, str(f),
                        "-o", str(s)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    return s.read_text()


def _body(asm, kernel):
    """The instructions of one kernel, comments and directives dropped (labels kept)."""
    m = re.search(rf"^{kernel}:.*?^\.Lfunc_end", asm, re.S | re.M)  # (blocks laid out behind s_endpgm belong to the kernel)
    assert m, kernel
    out = []
    for line in m.group(0).splitlines()[1:-1]:
        line = line.split(";")[0].strip()
        if line and not line.startswith((".", "//")) or re.match(r"\.LBB\d+_\d+:", line):
            out.append(" ".join(line.replace(",", ", ").split()))
    return out


_DPP_CTRL = re.compile(r"((?:quad_perm:\[[0-9, ]+\]|row_\w+:\d+|row_mirror|row_half_mirror|wave_\w+:\d+).*)$")


def _dpp(body):
    """[(opcode, control words)] of every DPP instruction, in program order."""
    out = []
    for ins in body:
        if "_dpp" in ins.split()[0]:
            m = _DPP_CTRL.search(ins)
            assert m, ins
            out.append((ins.split()[0], " ".join(m.group(1).replace(", ", ",").split())))
    return out


@pytest.fixture(scope="module")
def listings(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("needs hipcc")
    d = tmp_path_factory.mktemp("asm_blocks")
    return {"asm": _compile(d, "asm", CSRC), "llvm": _compile(d, "llvm", CSRC, defines=("NDZIP_NO_EXEC_ASM",))}


@pytest.mark.parametrize("D", [1, 2, 4, 8])
def test_row_scan_step64_takes_the_dpp_controls_llvm_takes(listings, D):
    want = f"row_shr:{D} row_mask:0xf bank_mask:0xf bound_ctrl:1"
    hand, llvm = _dpp(_body(listings["asm"], f"k_row{D}")), _dpp(_body(listings["llvm"], f"k_row{D}"))
    # one DPP read of the low and one of the high dword of each of the eight values, every one with the same control word
    assert [c for _, c in llvm] == [want] * 16, llvm
    assert [c for _, c in hand] == [want] * 16, hand
    assert [o for o, _ in hand] == ["v_add_co_u32_dpp", "v_addc_co_u32_dpp"] * 8
    assert {o for o, _ in llvm} <= {"v_mov_b32_dpp", "v_add_u32_dpp"}  # (LLVM folds a DPP move into a plain add, never into a carry add)
    _carry_pairs_are_padded(_body(listings["asm"], f"k_row{D}"))


def _carry_pairs_are_padded(body):
    """gfx940-family: a VALU instruction reads VCC written by a VALU instruction no earlier than 2 wait states later -- hipcc's own
    64-bit subtractions read v_sub_co_u32 / s_nop 1 / v_subb_co_u32; the hand-written pairs must read the same."""
    n = 0
    for i, ins in enumerate(body):
        if ins.startswith("v_add_co_u32_dpp"):
            assert body[i + 1] == "s_nop 1" and body[i + 2].startswith("v_addc_co_u32_dpp"), body[i:i + 3]
            lo, hi = ins.split()[1].rstrip(","), body[i + 2].split()[1].rstrip(",")
            assert re.fullmatch(r"v\d+", lo) and re.fullmatch(r"v\d+", hi) and lo != hi
            n += 1
    assert n > 0


def test_hipcc_pads_its_own_carry_pairs_the_same_way(tmp_path):
    """... and that IS how the compiler pads a carry pair it emits itself on gfx950 (a 64-bit subtraction under a borrow chain)."""
    if not os.path.exists(HIPCC):
        pytest.skip("needs hipcc")
    src = tmp_path / "sub.hip"
    src.write_text('#include <hip/hip_runtime.h>\n__global__ void k(const unsigned long long *a, unsigned long long *o) { o[threadIdx.x] = a[threadIdx.x] - a[threadIdx.x + 64]; }\n')
    out = tmp_path / "sub.s"
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-S", "--cuda-device-only", str(src), "-o", str(out)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    body = [" ".join(l.split(";")[0].split()) for l in out.read_text().splitlines() if l.strip() and not l.strip().startswith((".", ";"))]
    i = next(k for k, l in enumerate(body) if l.startswith("v_sub_co_u32"))
    assert body[i + 1] == "s_nop 1" and body[i + 2].startswith("v_subb_co_u32"), body[i:i + 3]


def test_wave_inclusive_scan64_is_the_32_bit_scan_with_the_carry_taken_along(listings):
    want = ["row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1", "row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1",
            "row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1", "row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1",
            "row_bcast:15 row_mask:0xa bank_mask:0xf", "row_bcast:31 row_mask:0xc bank_mask:0xf"]  # == LLVM's own wave64 scan (first test of this file)
    hand, llvm = _dpp(_body(listings["asm"], "k_scan64")), _dpp(_body(listings["llvm"], "k_scan64"))
    # per step one DPP read of the low and one of the high dword, both with the step's control word, steps in this order
    assert [c for _, c in hand] == [c for c in want for _ in (0, 1)], hand
    assert sorted(c for _, c in llvm) == sorted(c for c in want for _ in (0, 1)), llvm
    order = []
    for _, c in llvm:
        if c not in order:
            order.append(c)
    assert order == want
    assert [o for o, _ in hand] == ["v_add_co_u32_dpp", "v_addc_co_u32_dpp"] * 6
    _carry_pairs_are_padded(_body(listings["asm"], "k_scan64"))


def _symbolic_pair(body):
    """{output word: expression} of k_pair.  The straight-line subset the two listings use: v_and, v_cmp_{eq, ne} 0, DPP move /
    DPP select with quad_perm:[1,0,3,2] (the DPP operand is src0: VOP2 DPP modifies src0 only), v_cndmask (D = VCC ? src1 : src0)."""
    reg, vcc, out = {}, None, {}

    def rng(tok):
        m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
        return [f"v{i}" for i in range(int(m.group(1)), int(m.group(2)) + 1)] if m else [tok]

    def val(r):
        return reg.get(r, r)

    for ins in body:
        t = [x.rstrip(",") for x in ins.split()]
        op = t[0]
        off = int(re.search(r"offset:(\d+)", ins).group(1)) // 4 if "offset:" in ins else 0
        if op.startswith("global_load_dwordx4"):
            for k, r in enumerate(rng(t[1])):
                reg[r] = f"in{off + k}"
        elif op.startswith("global_store_dwordx4"):
            for k, r in enumerate(rng(t[2])):
                out[off + k] = val(r)
        elif op == "v_and_b32_e32" and t[2] == "1":
            reg[t[1]] = "odd" if val(t[3]) in ("v0", "tid") else f"and1({val(t[3])})"
        elif op in ("v_cmp_ne_u32_e32", "v_cmp_eq_u32_e32") and t[1] == "vcc" and t[2] == "0":
            vcc = (op[6:8], val(t[3]))
        elif op == "v_mov_b32_dpp":
            assert "quad_perm:[1, 0, 3, 2]" in ins, ins
            reg[t[1]] = f"swap({val(t[2])})"
        elif op in ("v_cndmask_b32_e32", "v_cndmask_b32_dpp"):
            assert t[4] == "vcc" and vcc is not None, ins
            s0, s1 = val(t[2]), val(t[3])
            if op.endswith("_dpp"):
                assert "quad_perm:[1, 0, 3, 2]" in ins, ins
                s0 = f"swap({s0})"
            picked_if_odd, otherwise = (s1, s0) if vcc[0] == "ne" else (s0, s1)
            assert vcc[1] == "odd", vcc
            reg[t[1]] = f"odd ? {picked_if_odd} : {otherwise}"
        elif op.startswith(("v_lshlrev", "s_", "v_mov_b32_e32")):
            if op == "v_mov_b32_e32":
                reg[t[1]] = val(t[2])
        else:
            raise AssertionError(f"instruction outside the subset: {ins}")
    return out


def test_pair_exchange_select4_selects_what_llvm_selects(listings):
    hand, llvm = _body(listings["asm"], "k_pair"), _body(listings["llvm"], "k_pair")
    ctrl = "quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"
    assert [(o, c) for o, c in _dpp(hand)] == [("v_cndmask_b32_dpp", ctrl)] * 8
    # (LLVM sets bound_ctrl on the move; a quad permutation under a full EXEC never reads an invalid lane, so the bit decides nothing)
    assert [(o, c.replace(" bound_ctrl:1", "")) for o, c in _dpp(llvm)] == [("v_mov_b32_dpp", ctrl)] * 8
    a, b = _symbolic_pair(hand), _symbolic_pair(llvm)
    want = {j: f"odd ? in{4 + j} : swap(in{j})" for j in range(4)}            # lo[j] = odd ? own b[j] : the other lane's a[j]
    want.update({4 + j: f"odd ? swap(in{4 + j}) : in{j}" for j in range(4)})  # hi[j] = odd ? the other lane's b[j] : own a[j]
    assert b == want, b   # what the C++ of the fallback says, as LLVM compiled it
    assert a == b, a      # ... and the hand-written v_cndmask_b32_dpp sequence computes the same eight words
    # VCC hand-off: v_cmp (VALU) -> v_cndmask reading VCC as its mask: 2 wait states on gfx940 / gfx950
    for i, ins in enumerate(hand):
        if ins.startswith("v_cmp_"):
            assert hand[i + 1] == "s_nop 1" and hand[i + 2].startswith("v_cndmask_b32_dpp"), hand[i:i + 3]


def _imm(tok):
    """An assembler immediate as its 32-bit pattern (inline float constants are printed as floats: 2.0 = 0x40000000)."""
    import struct

    if re.fullmatch(r"-?\d+\.\d+", tok):
        return struct.unpack("<I", struct.pack("<f", float(tok)))[0]
    return int(tok, 0) & 0xFFFFFFFF


def _load_positions(body):
    """{register: index of the word of the kernel's input it was loaded with} from the global_load_dword(x4) of the prologue."""
    pos = {}
    for ins in body:
        t = [x.rstrip(",") for x in ins.split()]
        if t[0] in ("global_load_dwordx4", "global_load_dword", "global_load_dwordx2", "global_load_dwordx3") and t[3].startswith("s["):
            off = int(re.search(r"offset:(\d+)", ins).group(1)) // 4 if "offset:" in ins else 0
            m = re.fullmatch(r"v\[(\d+):(\d+)\]", t[1])
            regs = [f"v{i}" for i in range(int(m.group(1)), int(m.group(2)) + 1)] if m else [t[1]]
            for k, r in enumerate(regs):
                pos[r] = off + k
    return pos


def test_lds_append_nonzero_stores_what_the_compiled_loop_stores(listings):
    hand, llvm = _body(listings["asm"], "k_append"), _body(listings["llvm"], "k_append")
    # hand: 32 x (v_cmpx_ne_u32 0 != w ; ds_write_b32 a, w ; v_add_u32 a += 4 ; s_mov_b64 exec, saved), same w in compare and store, one running `a`
    pos, seq, addr = _load_positions(hand), [], set()
    for i, ins in enumerate(hand):
        if ins.startswith("v_cmpx_"):
            t = [x.rstrip(",") for x in ins.split()]
            assert t[:3] == ["v_cmpx_ne_u32_e32", "vcc", "0"], ins
            w = t[3]
            st, ad, mv = hand[i + 1].replace(",", "").split(), hand[i + 2].replace(",", "").split(), hand[i + 3].replace(",", "").split()
            assert st[0] == "ds_write_b32" and st[2] == w, hand[i:i + 4]
            assert ad[:3] == ["v_add_u32_e32", st[1], "4"] and ad[3] == st[1], hand[i:i + 4]
            assert mv[:2] == ["s_mov_b64", "exec"], hand[i:i + 4]
            seq.append(pos[w])
            addr.add(st[1])
    assert seq == list(range(32)) and len(addr) == 1, (seq, addr)
    # llvm: per word v_cmp_ne_u32 0 != w -> s_and_saveexec -> (in the guarded block) ds_write_b32 a, w ; a += 4
    pos, tested, stored = _load_positions(llvm), [], []
    for i, ins in enumerate(llvm):
        t = [x.rstrip(",") for x in ins.split()]
        if t[0] == "v_cmp_ne_u32_e32" and t[1:3] == ["vcc", "0"] and t[3] in pos and llvm[i + 1].startswith("s_and_saveexec_b64"):
            tested.append(pos[t[3]])
        if t[0] == "ds_write_b32" and t[2] in pos:
            stored.append(pos[t[2]])
            near = " ".join(llvm[max(0, i - 3):i + 3])
            assert re.search(r"v_add_u32_e32 v\d+, 4, v\d+|v_add3_u32 v\d+, 0, 4, v\d+", near), near  # the address advances by one word
    # (a word is tested once on the fall-through path and once more in the guarded block of its predecessor: each of the 32 at least once)
    assert sorted(set(tested)) == list(range(32)) and sorted(stored) == list(range(32)), (tested, stored)


def test_lds_append_flagged64_addresses_and_tests_what_the_compiled_loop_does(listings):
    hand, llvm = _body(listings["asm"], "k_append64"), _body(listings["llvm"], "k_append64")

    def swizzles(body):
        """[(shift source, bitop3 operands, truth table, value of the scalar operand)] of every v_bitop3_b32 of the kernel"""
        sreg, shr, out = {}, {}, []
        for ins in body:
            t = [x.rstrip(",") for x in ins.split()]
            if t[0] in ("s_movk_i32", "s_mov_b32") and re.fullmatch(r"s\d+", t[1]):
                sreg[t[1]] = _imm(t[2])
            elif t[0] == "v_lshrrev_b32_e32":
                shr[t[1]] = (_imm(t[2]), t[3])
            elif t[0] == "v_bitop3_b32":
                table = re.search(r"bitop3:(0x[0-9a-f]+)", ins).group(1)
                out.append((shr.get(t[2]), t[3], table, sreg.get(t[4])))
        return out

    # at(a) = a ^ ((a >> 3) & 0x70): (a >> 3) in src0, a in src1, 0x70 from an SGPR in src2, truth table 0x6c -- in BOTH listings
    for name, body in (("hand", hand), ("llvm", llvm)):
        sw = swizzles(body)
        assert len(sw) >= 31, (name, len(sw))  # (LLVM folds the very first one, whose `a` is a known multiple of 256, into the address arithmetic or keeps it: 31 or 32)
        for shift, a, table, scalar in sw:
            assert table == "0x6c" and scalar == 0x70, (name, shift, a, table, scalar)
            assert shift is not None and shift[0] == 3 and shift[1] == a, (name, shift, a)
    assert len(swizzles(hand)) == 32
    # hand: 32 x (v_cmpx_gt_i32 0 > flags ; shift ; swizzle ; ds_write_b32 at, w ; a += 8 ; exec back ; flags <<= 1): plane i is kept where bit 31 - i is set
    pos, seq = _load_positions(hand), []
    for i, ins in enumerate(hand):
        if ins.startswith("v_cmpx_"):
            t = [x.rstrip(",") for x in ins.split()]
            assert t[:3] == ["v_cmpx_gt_i32_e32", "vcc", "0"], ins
            f = t[3]
            blk = [x.replace(",", "").split() for x in hand[i + 1:i + 7]]
            assert [b[0] for b in blk] == ["v_lshrrev_b32_e32", "v_bitop3_b32", "ds_write_b32", "v_add_u32_e32", "s_mov_b64", "v_lshlrev_b32_e32"], hand[i:i + 7]
            assert blk[2][1] == blk[1][1] and blk[3][1:] == [blk[0][3], "8", blk[0][3]] and blk[4][1] == "exec" and blk[5][1:] == [f, "1", f], hand[i:i + 7]
            seq.append(pos[blk[2][2]])
    assert seq == list(range(32)), seq
    # llvm: the same planes in the same order, tested by the masks 1 << 31 (as the sign: v_cmp_gt_i32 0 > flags, the hand-written test) .. 1 << 0
    masks, flag_reg = [], None
    for ins in llvm:
        t = [x.rstrip(",") for x in ins.split()]
        if t[0] == "v_cmp_gt_i32_e32" and t[1:3] == ["vcc", "0"]:
            flag_reg = t[3]
            masks.append(1 << 31)
        elif t[0] == "v_and_b32_e32" and flag_reg is not None and t[3] == flag_reg:
            masks.append(_imm(t[2]))
    assert masks == [1 << (31 - i) for i in range(32)], [hex(m) for m in masks]
    pos = _load_positions(llvm)
    stored = [pos[t.replace(",", "").split()[2]] for t in llvm if t.startswith("ds_write_b32") and t.replace(",", "").split()[2] in pos]
    assert stored == [i for i in range(32) if i in stored] and len(stored) >= 31, stored  # (the first plane's word may be re-loaded under the branch)


def test_a_changed_control_word_is_caught(tmp_path):
    """The point of the exercise: ONE token of ONE control word changed in a copy of gfx950_lds.hpp (a row mask of the 64-bit scan, the
    quad permutation of the pair exchange, the truth table of the swizzle, the compared constant of the compaction) and the comparison
    with LLVM's lowering of the untouched fallback fails."""
    if not os.path.exists(HIPCC):
        pytest.skip("needs hipcc")
    src = open(os.path.join(CSRC, "gfx950_lds.hpp")).read()
    llvm, clean = _compile(tmp_path, "llvm", CSRC, defines=("NDZIP_NO_EXEC_ASM",)), _compile(tmp_path, "asm", CSRC)
    edits = {
        "scan64 row mask": ('NDZIP_SCAN64_STEP("row_bcast:15 row_mask:0xa bank_mask:0xf")', 'NDZIP_SCAN64_STEP("row_bcast:15 row_mask:0xc bank_mask:0xf")',
                            lambda L: test_wave_inclusive_scan64_is_the_32_bit_scan_with_the_carry_taken_along(L)),
        "row step bound_ctrl": ('row_shr:" #d " row_mask:0xf bank_mask:0xf bound_ctrl:1\\n\\ts_nop 1', 'row_shr:" #d " row_mask:0xf bank_mask:0xf\\n\\ts_nop 1',
                                lambda L: test_row_scan_step64_takes_the_dpp_controls_llvm_takes(L, 4)),
        "pair quad_perm": ('%[b" #n "], vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\\n\\t"\n#define NDZIP_SWAPSEL_HI', '%[b" #n "], vcc quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\\n\\t"\n#define NDZIP_SWAPSEL_HI',
                           lambda L: test_pair_exchange_select4_selects_what_llvm_selects(L)),
        "pair operand order": ('"v_cndmask_b32_dpp %[hi" #n "], %[b" #n "], %[a" #n "], vcc', '"v_cndmask_b32_dpp %[hi" #n "], %[a" #n "], %[b" #n "], vcc',
                               lambda L: test_pair_exchange_select4_selects_what_llvm_selects(L)),
        "swizzle truth table": ("bitop3:0x6c", "bitop3:0x78", lambda L: test_lds_append_flagged64_addresses_and_tests_what_the_compiled_loop_does(L)),
        "append compare": ('"v_cmpx_ne_u32_e32 vcc, 0, %[w" #n "]', '"v_cmpx_ne_u32_e32 vcc, 1, %[w" #n "]', lambda L: test_lds_append_nonzero_stores_what_the_compiled_loop_stores(L)),
    }
    for what, (old, new, check) in edits.items():
        assert src.count(old) == 1, (what, src.count(old))
        d = tmp_path / what.replace(" ", "_")
        d.mkdir()
        (d / "gfx950_lds.hpp").write_text(src.replace(old, new))
        hand = _compile(d, "asm", str(d))
        with pytest.raises(AssertionError):
            check({"asm": hand, "llvm": llvm})
        check({"asm": clean, "llvm": llvm})  # (and the untouched header passes the very same call)
