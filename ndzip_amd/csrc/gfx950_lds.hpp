// ndzip_amd/csrc/gfx950_lds.hpp -- the few primitives that have to be spelled in gfx950 terms: a 16-byte-per-lane LDS read
// whose address is pinned through a VGPR, the wait for a wavefront's outstanding vector-memory operations, and the
// read-once (non-temporal) 16-byte global load.  Kept in their own header so that the wave64 functional model used by the
// CPU test suite (tests/wavesim, test infrastructure only) can compile every other line of the kernels unchanged and
// substitute just this file.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

namespace ndzip_hip {

#define NDZIP_DEV __device__ __forceinline__

struct alignas(16) vec16 {
    uint32_t w[4];
};

// 16 bytes per lane from LDS in one ds_read_b128.  Ground truth on gfx950 (tools/ldsbench2.hip, inline-asm reads,
// SQ_LDS_IDX_ACTIVE per wave-instruction): ds_read_b128 at 16-byte-aligned lane strides of 16 / 144 / 272 bytes = 4
// (the peak: 16 lanes x 16 bytes per count), the equivalent ds_read2_b64 = 16, ds_read_b128 at an address that is
// only 8-byte aligned = 64.  A b128 access is served in groups of 16 consecutive lanes over sixteen 16-byte slots
// (address / 16 mod 16); two lanes of a group in the same slot at different addresses double the group's cost.
NDZIP_DEV vec16 lds_read16(const char *p) {
    // The address-space cast keeps it a ds_ access.  The address is NOT pinned through a register (rounds 1-2 passed it through an
    // empty volatile asm to keep hipcc from hoisting all 24 stencil reads to the top): every pinned read needed its own
    // address VGPR -- base + 16, base + 32 ... as 22 loop-invariant registers plus a v_mov per read -- where the unpinned read
    // takes the constant in its offset field.  The order of the reads is held by the scheduling barriers between the groups of
    // the stencil instead (compress_kernel_db<float, 3>: 157 -> 138 VGPRs, 758 -> 729 VALU instructions per iteration, no scratch).
    using lds_char = const __attribute__((address_space(3))) char;
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    using lds_vec = const __attribute__((address_space(3))) u32x4;
    const u32x4 q = *reinterpret_cast<lds_vec *>((lds_char *) p);
    vec16 v;
    v.w[0] = q.x;
    v.w[1] = q.y;
    v.w[2] = q.z;
    v.w[3] = q.w;
    return v;
}

// An LDS location as its 32-bit byte address and back (what a ds_ instruction takes): address arithmetic that is more than
// pointer + offset -- the XOR swizzle of run_layout -- is done on this integer, so that it stays 32-bit VALU work and the
// access stays a ds_ instruction (an integer round trip through a 64-bit generic pointer would turn it into a flat access).
NDZIP_DEV uint32_t lds_address(const void *p) {
    using lds_char = const __attribute__((address_space(3))) char;
    return static_cast<uint32_t>(reinterpret_cast<uintptr_t>((lds_char *) p));
}
NDZIP_DEV char *lds_pointer(uint32_t address) {
    using lds_char = __attribute__((address_space(3))) char;
    return (char *) reinterpret_cast<lds_char *>(static_cast<uintptr_t>(address));
}

// A value that is the same in every lane of the wavefront (a wave index, a hypercube-of-the-tile index), as a scalar: what is
// tested on it becomes an s_cmp and a scalar branch instead of a v_cmp into a 64-bit lane mask that has to be kept (or spilled
// and reloaded lane by lane) for as long as the condition is used.
//
// Bisecting aids (never defined in the product build; ndzip_amd/build.py builds ndzip_amd/_variants/plain.so with both for
// tools/variant_parity.py): -DNDZIP_NO_SCALAR_PINS takes every uniformity claim back -- wave_uniform / scalar_pointer return the
// lane's own value, all addressing falls back to per-lane 64-bit pointers -- and -DNDZIP_NO_EXEC_ASM replaces EVERY block of
// instruction-level assembly in this header by compiled code that means the same: the EXEC-masked store sequences of
// lds_append_nonzero / lds_append_flagged64 by the loops they stand for, and the carry-chained / select-fused DPP sequences of
// row_scan_step64, wave_inclusive_scan64 and pair_exchange_select4 by __builtin_amdgcn_update_dpp moves plus ordinary 64-bit adds and
// selects (the compiler then counts the wait states itself).  If the product library ever disagrees with the oracle on hardware and
// the plain variant does not, the fault is in one of these mechanisms and not in the algorithm; tools/variant_parity.py tells which
// profile (f32: append / pins; f64: also the three DPP helpers).
#ifdef NDZIP_NO_SCALAR_PINS
NDZIP_DEV int wave_uniform(int x) { return x; }
#else
NDZIP_DEV int wave_uniform(int x) { return __builtin_amdgcn_readfirstlane(x); }
#endif

// The same value, but nothing derived from it is loop-invariant as far as the optimiser can tell: predicates and wave indices
// computed from this copy of the work-item id are re-derived (a compare, a v_readfirstlane + shift) where they are used instead of
// being hoisted out of the persistent loop as 64-bit lane masks and scalars -- of which the compress kernels had more than the
// SGPR file holds (34-47 spilled to VGPR lanes, read back with v_readlane + hazard nops every iteration).  Emits nothing.
NDZIP_DEV int fresh_copy(int x) {
    asm volatile("" : "+v"(x));
    return x;
}

// A wave-uniform 64-bit address, pinned in an SGPR pair: `pointer + (uint32_t) lane_offset` behind it is selected as the
// SGPR-base + 32-bit-VGPR-offset form of global_load / global_store (one VGPR of address per lane instead of two, no 64-bit VALU
// additions).  Without the pin LLVM re-associates (uniform + uniform) + lane into (uniform + lane) + uniform and is back at a
// per-lane 64-bit pointer.
// (`p` points to GLOBAL memory: the integer round trip would otherwise leave a generic pointer, i.e. flat_ accesses.)
template<typename P>
NDZIP_DEV P *scalar_pointer(P *p) {
#ifdef NDZIP_NO_SCALAR_PINS
    return p;
#endif
    // (v_readfirstlane folds away when the compiler can prove the value uniform, and makes the claim true where it cannot: an
    // "s" operand fed from a VGPR is a back-end error, not a copy)
    using global_p = __attribute__((address_space(1))) P;
    const unsigned long long v = reinterpret_cast<unsigned long long>(p);
    unsigned long long a = static_cast<unsigned long long>(static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(v))))
            | (static_cast<unsigned long long>(static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(v >> 32)))) << 32);
    asm("" : "+s"(a));
    return (P *) reinterpret_cast<global_p *>(a);
}

// The 32-bit per-lane byte offset that goes with a scalar_pointer, re-materialised where it is used: instruction selection works
// one basic block at a time, and only sees "SGPR pair + zero-extended 32-bit VGPR" (the SGPR-base addressing form) if the
// zero-extension happens in the block of the access -- hoisted out of the loop it is just some 64-bit VGPR.  Emits nothing.
NDZIP_DEV uint32_t lane_offset_here(uint32_t bytes) {
    asm volatile("" : "+v"(bytes));
    return bytes;
}

// Scheduling fence for a batch of 32 word reads from LDS: every LDS access above it is issued before any of `w` is used
// below it.  Left alone, hipcc interleaves a dependent-address gather with the uses of its results (ds_read_b32 ;
// s_waitcnt lgkmcnt(0) ; v_and ; next address ; ds_read_b32 ...), one exposed LDS round trip per word; with the fence the
// address chain and the reads go out back to back and the wavefront waits once.  (Two statements: an asm takes at most 30
// operands.  The "memory" clobber is what keeps the reads above the first one.)
NDZIP_DEV void lds_reads_issued_before_use(uint32_t (&w)[32]) {
    asm volatile("" : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]), "+v"(w[4]), "+v"(w[5]), "+v"(w[6]), "+v"(w[7]), "+v"(w[8]),
            "+v"(w[9]), "+v"(w[10]), "+v"(w[11]), "+v"(w[12]), "+v"(w[13]), "+v"(w[14]), "+v"(w[15])::"memory");
    asm volatile("" : "+v"(w[16]), "+v"(w[17]), "+v"(w[18]), "+v"(w[19]), "+v"(w[20]), "+v"(w[21]), "+v"(w[22]), "+v"(w[23]),
            "+v"(w[24]), "+v"(w[25]), "+v"(w[26]), "+v"(w[27]), "+v"(w[28]), "+v"(w[29]), "+v"(w[30]), "+v"(w[31])::"memory");
}

// Ordering pin for 32 register values: everything that computes w[0 .. 32) comes before this point, no memory access of the
// program crosses it (the "memory" clobber), nothing is emitted.  Used where the compiler's own placement put a wait for ALL
// outstanding vector-memory operations in front of a long stretch of register-only work (compress kernels: the 32x32 transposes
// behind the copy-out's stores; see codec_launch.inl).  Two statements: an asm takes at most 30 operands.
NDZIP_DEV void registers_complete_here(uint32_t (&w)[32]) {
    asm volatile("" : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]), "+v"(w[4]), "+v"(w[5]), "+v"(w[6]), "+v"(w[7]), "+v"(w[8]),
            "+v"(w[9]), "+v"(w[10]), "+v"(w[11]), "+v"(w[12]), "+v"(w[13]), "+v"(w[14]), "+v"(w[15])::"memory");
    asm volatile("" : "+v"(w[16]), "+v"(w[17]), "+v"(w[18]), "+v"(w[19]), "+v"(w[20]), "+v"(w[21]), "+v"(w[22]), "+v"(w[23]),
            "+v"(w[24]), "+v"(w[25]), "+v"(w[26]), "+v"(w[27]), "+v"(w[28]), "+v"(w[29]), "+v"(w[30]), "+v"(w[31])::"memory");
}

// The value, as far as the optimiser is concerned, from nowhere: stops hipcc from rewriting `p + 8 * (x << i >> 31)` into
// bfe(4 bits) / and -8 / add (three instructions) where v_bfe_i32 + v_lshl_add_u32 (two) do, and lets the 0 / -1 mask be
// reused for the select afterwards.
NDZIP_DEV int32_t opaque_vgpr(int32_t x) {
    asm("" : "+v"(x));
    return x;
}

// All vector-memory operations this wavefront has issued (loads, stores, atomics -- gfx9 counts them in one counter) have
// completed.  Between write-through / atomic accesses this is all the ordering an agent-scope hand-off needs.
NDZIP_DEV void wait_for_own_memory_operations() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// Agent-scope exchange / OR that returns only once it HAS BEEN PERFORMED: the returning form of the atomic (executed where
// every XCD's agent-scope accesses meet, not in this XCD's L2) whose result the wavefront has received -- the empty asm takes
// the result in a VGPR, so the compiler has to wait for it.  This is what the launch's last hand-offs are built from (error
// word and stream length before `workgroups done`): vmcnt alone says a plain write-through store has reached THIS XCD's L2,
// which is not a statement about what a workgroup on another XCD observes; a returned atomic is.  Costs one memory round
// trip per workgroup, once, on its way out (an agent-scope release fence = write-back of the XCD's L2 would order it too,
// at several microseconds per workgroup).
NDZIP_DEV uint32_t exchange_performed(uint32_t *p, uint32_t v) {
    uint32_t old = __hip_atomic_exchange(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("" : "+v"(old)::"memory");
    return old;
}
NDZIP_DEV uint32_t fetch_or_performed(uint32_t *p, uint32_t v) {
    uint32_t old = __hip_atomic_fetch_or(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("" : "+v"(old)::"memory");
    return old;
}

// 16 bytes from a 16-byte aligned global address that THIS launch reads exactly once and no other workgroup needs from the
// same cache line: global_load_dwordx4 ... nt.  The MI355X guide measures read-once streams with the nt policy at 6.5-6.8
// instead of 6.4 TB/s chip-wide and 18-19 % less issue-to-landed latency (nothing useful is kept in, or evicted from, L2).
NDZIP_DEV vec16 global_load16_once(const void *p) {
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 q = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(p));
    vec16 v;
    v.w[0] = q.x;
    v.w[1] = q.y;
    v.w[2] = q.z;
    v.w[3] = q.w;
    return v;
}

// The 16-byte aligned block around words of a stream, read once.  The decoder fetches an encoded run as whole aligned blocks,
// so a run's first and last block can hold up to three words of the neighbouring runs -- or, at the two ends of a stream, of
// nobody.  An aligned 16-byte load cannot straddle a page: the access is safe whenever one of its words is, and the surplus
// words are never looked at.  (The functional model checks exactly this contract under AddressSanitizer: at least one word of
// the block inside the caller's buffer, junk substituted for the others.)
NDZIP_DEV vec16 global_load16_block(const void *p) { return global_load16_once(p); }

// Compaction of a chunk: the NON-ZERO words of w[0 .. 32) stored back to back from LDS byte address `a` on, in order; returns
// the address behind the last one.  Per word: v_cmpx_ne_u32 (the lanes whose word is non-zero stay active), ds_write_b32,
// v_add_u32 of the running address (under the same mask), s_mov_b64 exec back to the mask the sequence was entered with --
// two VALU, one LDS and one SALU instruction and NO branch.  Written as `if (w[i]) *p++ = w[i]`, hipcc emits v_cmp, s_and_saveexec,
// s_cbranch_execz, the store, the increment and s_or_b64 exec per word: 32 x (4 scalar instructions, one of them a branch, and a
// saved 64-bit mask that lives in SGPRs) -- 250 of the 600 scalar-side instructions of a compress iteration, and enough SGPR
// pressure that the kernel spilled 49 SGPRs to VGPR lanes (v_writelane / v_readlane in the loop).
// The asm leaves EXEC as it found it at every statement boundary (the compiler does not know EXEC was touched and need not);
// VCC is clobbered.  Hazards (gfx9 "manually inserted wait states"): none between a VALU write of EXEC and an LDS instruction
// or a VALU instruction that is not DPP; a DPP instruction needs 5 wait states after v_cmpx -- the last v_cmpx is followed by
// three instructions and s_nop 1 before control returns to compiled code.
#define NDZIP_APPEND1(n) \
    "v_cmpx_ne_u32_e32 vcc, 0, %[w" #n "]\n\tds_write_b32 %[a], %[w" #n "]\n\tv_add_u32_e32 %[a], 4, %[a]\n\ts_mov_b64 exec, %[full]\n\t"
#define NDZIP_APPEND8 NDZIP_APPEND1(0) NDZIP_APPEND1(1) NDZIP_APPEND1(2) NDZIP_APPEND1(3) NDZIP_APPEND1(4) NDZIP_APPEND1(5) NDZIP_APPEND1(6) NDZIP_APPEND1(7)
NDZIP_DEV uint32_t lds_append_nonzero(uint32_t a, const uint32_t (&w)[32]) {
#ifdef NDZIP_NO_EXEC_ASM
#pragma unroll
    for (int i = 0; i < 32; ++i) {
        if (w[i] != 0) {
            *reinterpret_cast<uint32_t *>(lds_pointer(a)) = w[i];
            a += 4;
        }
    }
    return a;
#endif
    unsigned long long full;
    asm volatile("s_mov_b64 %0, exec" : "=s"(full));
#pragma unroll
    for (int i = 0; i < 32; i += 8) {
        asm volatile(NDZIP_APPEND8 "s_nop 1"
                     : [a] "+v"(a)
                     : [w0] "v"(w[i]), [w1] "v"(w[i + 1]), [w2] "v"(w[i + 2]), [w3] "v"(w[i + 3]), [w4] "v"(w[i + 4]), [w5] "v"(w[i + 5]),
                     [w6] "v"(w[i + 6]), [w7] "v"(w[i + 7]), [full] "s"(full)
                     : "vcc", "memory");
    }
    return a;
}
#undef NDZIP_APPEND8
#undef NDZIP_APPEND1

// The same for the 64-bit profiles' encoder (codec_kernels_wide.hpp: a lane holds ONE dword of each of 32 plane words, the other
// dword is another lane's): word w[i] is kept where bit 31 - i of `flags` (the chunk head's half for this lane's planes) is set --
// a plane's dword may be zero while the plane is not -- and goes to LDS byte address at(a) = a ^ ((a >> 3) & 0x70), the XOR swizzle
// of run_layout<uint64_t>, `a` advancing by 8 (one 64-bit stream word) per kept plane.  Per plane: v_cmpx_gt_i32 0 > flags (the
// lanes whose flag MSB is set stay active), v_lshrrev + v_bitop3 (the swizzle, 0x70 from an SGPR), ds_write_b32, v_add_u32 under
// the mask, s_mov_b64 exec back, v_lshlrev flags << 1 in all lanes: five VALU, one LDS, one SALU instruction.  Compiled from
// `if (flags >> (31 - i) & 1) { *at(a) = w[i]; a += 8; }` it is v_and + v_cmp + s_and_saveexec + the same four + s_or_b64: the same
// VALU count, TWO scalar instructions and a saved 64-bit mask per plane (round 4: 177 s_and_saveexec + 205 s_or_b64 executed per
// hypercube, SGPR spills in all three compress_kernel_wide).
// Wait states (checked by tools/asm_hazards.py against LLVM's gfx950 hazard recogniser): none between v_cmpx and ds_write / a
// non-DPP VALU instruction; the SGPR operands are written by SALU instructions (s_mov), which a VALU instruction may read at once;
// behind the last v_cmpx come six instructions (a DPP instruction needs 5 wait states after a VALU write of EXEC) and the closing
// s_nop 1 covers a DPP read of `a` / `flags` by compiled code.  EXEC is as found at every statement boundary; VCC is clobbered.
#define NDZIP_APPEND64_1(n) \
    "v_cmpx_gt_i32_e32 vcc, 0, %[f]\n\tv_lshrrev_b32_e32 %[t], 3, %[a]\n\tv_bitop3_b32 %[t], %[t], %[a], %[m] bitop3:0x6c\n\t" \
    "ds_write_b32 %[t], %[w" #n "]\n\tv_add_u32_e32 %[a], 8, %[a]\n\ts_mov_b64 exec, %[full]\n\tv_lshlrev_b32_e32 %[f], 1, %[f]\n\t"
#define NDZIP_APPEND64_8 \
    NDZIP_APPEND64_1(0) NDZIP_APPEND64_1(1) NDZIP_APPEND64_1(2) NDZIP_APPEND64_1(3) NDZIP_APPEND64_1(4) NDZIP_APPEND64_1(5) NDZIP_APPEND64_1(6) NDZIP_APPEND64_1(7)
NDZIP_DEV void lds_append_flagged64(uint32_t a, uint32_t flags, const uint32_t (&w)[32]) {
#ifdef NDZIP_NO_EXEC_ASM
#pragma unroll
    for (int i = 0; i < 32; ++i) {
        if ((flags >> (31 - i)) & 1u) {
            *reinterpret_cast<uint32_t *>(lds_pointer(a ^ ((a >> 3) & 0x70u))) = w[i];
            a += 8;
        }
    }
    return;
#endif
    unsigned long long full;
    asm volatile("s_mov_b64 %0, exec" : "=s"(full));
    const uint32_t slot_bits = 0x70u;
    uint32_t t;
#pragma unroll
    for (int i = 0; i < 32; i += 8) {
        asm volatile(NDZIP_APPEND64_8 "s_nop 1"
                     : [a] "+v"(a), [f] "+v"(flags), [t] "=&v"(t)
                     : [w0] "v"(w[i]), [w1] "v"(w[i + 1]), [w2] "v"(w[i + 2]), [w3] "v"(w[i + 3]), [w4] "v"(w[i + 4]), [w5] "v"(w[i + 5]),
                     [w6] "v"(w[i + 6]), [w7] "v"(w[i + 7]), [full] "s"(full), [m] "s"(slot_bits)
                     : "vcc", "memory");
    }
}
#undef NDZIP_APPEND64_8
#undef NDZIP_APPEND64_1

// One step of a prefix sum over the 16 lanes of a DPP row for EIGHT 64-bit values held as (lo, hi) register pairs:
// v += row_shr:D(v), lanes shifted in from outside the row contributing 0.  In C++ this is two v_mov_b32_dpp and a 64-bit add per
// value (the DPP move folds into a plain v_add_u32, not into an add that produces or consumes a carry): 3 VALU instructions; here
// v_add_co_u32_dpp + v_addc_co_u32_dpp: 2 VALU instructions and two idle issue slots of this wavefront (which the SIMD's other
// wavefronts can use).  (f64 3D decode: 64 value-steps per work-item.)
// Wait states (the compiler does not look into an asm statement; tools/asm_hazards.py runs every statement of the built code
// through LLVM's own gfx950 hazard recogniser and fails if that would insert a single s_nop more):
//   * a DPP instruction must not read a VGPR a VALU instruction wrote less than 2 wait states earlier -- each statement opens and
//     closes with s_nop 1 (compiled code on either side; hipcc pads ONE state behind an asm statement, not two);
//   * gfx940 / gfx950: a VALU instruction must not read an SGPR or VCC a VALU instruction wrote less than 2 wait states earlier --
//     the carry of v_add_co is such a value: s_nop 1 between the two halves of every add (hipcc's own 64-bit subtractions read
//     v_sub_co / s_nop 1 / v_subb_co for the same reason).  Rounds 3-4 had the pair back to back: per LLVM a stale-carry hazard.
// VCC is clobbered.
#define NDZIP_ROWADD1(n, d) \
    "v_add_co_u32_dpp %[l" #n "], vcc, %[l" #n "], %[l" #n "] row_shr:" #d " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\ts_nop 1\n\t" \
    "v_addc_co_u32_dpp %[h" #n "], vcc, %[h" #n "], %[h" #n "], vcc row_shr:" #d " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
#define NDZIP_ROWADD8(d) "s_nop 1\n\t" NDZIP_ROWADD1(0, d) NDZIP_ROWADD1(1, d) NDZIP_ROWADD1(2, d) NDZIP_ROWADD1(3, d) NDZIP_ROWADD1(4, d) NDZIP_ROWADD1(5, d) NDZIP_ROWADD1(6, d) NDZIP_ROWADD1(7, d) "s_nop 1" 
#define NDZIP_ROWADD_OPERANDS \
    [l0] "+v"(lo[0]), [h0] "+v"(hi[0]), [l1] "+v"(lo[1]), [h1] "+v"(hi[1]), [l2] "+v"(lo[2]), [h2] "+v"(hi[2]), [l3] "+v"(lo[3]), [h3] "+v"(hi[3]), \
    [l4] "+v"(lo[4]), [h4] "+v"(hi[4]), [l5] "+v"(lo[5]), [h5] "+v"(hi[5]), [l6] "+v"(lo[6]), [h6] "+v"(hi[6]), [l7] "+v"(lo[7]), [h7] "+v"(hi[7])
template<int D>
NDZIP_DEV void row_scan_step64(uint32_t (&lo)[8], uint32_t (&hi)[8]) {
    static_assert(D == 1 || D == 2 || D == 4 || D == 8);
#ifdef NDZIP_NO_EXEC_ASM
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const uint32_t sl = static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(lo[j]), 0x110 + D, 0xf, 0xf, true));
        const uint32_t sh = static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(hi[j]), 0x110 + D, 0xf, 0xf, true));
        const uint64_t r = ((static_cast<uint64_t>(hi[j]) << 32) | lo[j]) + ((static_cast<uint64_t>(sh) << 32) | sl);
        lo[j] = static_cast<uint32_t>(r);
        hi[j] = static_cast<uint32_t>(r >> 32);
    }
    return;
#endif
    if constexpr (D == 1) {
        asm volatile(NDZIP_ROWADD8(1) : NDZIP_ROWADD_OPERANDS : : "vcc");
    } else if constexpr (D == 2) {
        asm volatile(NDZIP_ROWADD8(2) : NDZIP_ROWADD_OPERANDS : : "vcc");
    } else if constexpr (D == 4) {
        asm volatile(NDZIP_ROWADD8(4) : NDZIP_ROWADD_OPERANDS : : "vcc");
    } else {
        asm volatile(NDZIP_ROWADD8(8) : NDZIP_ROWADD_OPERANDS : : "vcc");
    }
}
#undef NDZIP_ROWADD_OPERANDS
#undef NDZIP_ROWADD8
#undef NDZIP_ROWADD1

// Inclusive prefix sum of ONE 64-bit value per lane over the 64 lanes of the wavefront, as (lo, hi): the six steps of the 32-bit
// DPP scan (row_shr 1 / 2 / 4 / 8, row_bcast:15 into rows 1 and 3, row_bcast:31 into rows 2 and 3) with the carry taken along --
// v_add_co_u32_dpp + v_addc_co_u32_dpp per step, twelve VALU instructions and no LDS traffic.  (__shfl_up on a 64-bit value is two
// ds_bpermute_b32 plus a compare, two selects and a 64-bit add per step: six dependent LDS-crossbar round trips and ~50
// instructions on the path of the 1D decoders' carry.)  Wait states (see row_scan_step64): s_nop 1 between add_co and addc for the
// carry in VCC; that also puts 3 wait states between a step's write of lo / hi and the next step's DPP read of it; s_nop 1 in
// front for whatever the compiler computed last and behind for whatever it reads first.
// Lanes of rows a row_mask disables are not written (their VCC bit is irrelevant: the addc is disabled too).
#define NDZIP_SCAN64_STEP(ctrl) \
    "v_add_co_u32_dpp %[lo], vcc, %[lo], %[lo] " ctrl "\n\ts_nop 1\n\tv_addc_co_u32_dpp %[hi], vcc, %[hi], %[hi], vcc " ctrl "\n\t"
NDZIP_DEV void wave_inclusive_scan64(uint32_t &lo, uint32_t &hi) {
#ifdef NDZIP_NO_EXEC_ASM
    {
        // (DPP control words: 0x110 + d = row_shr:d, 0x142 = row_bcast:15, 0x143 = row_bcast:31; lanes a row_mask disables take old = 0)
#define NDZIP_SCAN64_C(ctrl, rows, bound) \
    { \
        const uint32_t sl = static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(lo), ctrl, rows, 0xf, bound)); \
        const uint32_t sh = static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(hi), ctrl, rows, 0xf, bound)); \
        const uint64_t r = ((static_cast<uint64_t>(hi) << 32) | lo) + ((static_cast<uint64_t>(sh) << 32) | sl); \
        lo = static_cast<uint32_t>(r); \
        hi = static_cast<uint32_t>(r >> 32); \
    }
        NDZIP_SCAN64_C(0x111, 0xf, true)
        NDZIP_SCAN64_C(0x112, 0xf, true)
        NDZIP_SCAN64_C(0x114, 0xf, true)
        NDZIP_SCAN64_C(0x118, 0xf, true)
        NDZIP_SCAN64_C(0x142, 0xa, false)
        NDZIP_SCAN64_C(0x143, 0xc, false)
#undef NDZIP_SCAN64_C
        return;
    }
#endif
    asm volatile("s_nop 1\n\t"
                 NDZIP_SCAN64_STEP("row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1")
                 NDZIP_SCAN64_STEP("row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1")
                 NDZIP_SCAN64_STEP("row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1")
                 NDZIP_SCAN64_STEP("row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1")
                 NDZIP_SCAN64_STEP("row_bcast:15 row_mask:0xa bank_mask:0xf")
                 NDZIP_SCAN64_STEP("row_bcast:31 row_mask:0xc bank_mask:0xf")
                 "s_nop 1"
                 : [lo] "+v"(lo), [hi] "+v"(hi)
                 :
                 : "vcc");
}
#undef NDZIP_SCAN64_STEP

// Exchange between the two lanes of a pair (t, t ^ 1) fused with the choice of what to keep, for four word pairs (a[j], b[j]):
//   lo[j] = odd lane ? own b[j] : the other lane's a[j]        hi[j] = odd lane ? the other lane's b[j] : own a[j]
// as ONE v_cndmask_b32_dpp each (quad_perm [1,0,3,2] on the swapped operand) instead of a DPP move and a select: the compiler
// folds a DPP move into src0 of a VOP2 instruction, but not across the select's operand order / inverted mask it would take here.
// `odd_flag`: non-zero in odd lanes.  Every lane of the wavefront executes this (a DPP operand reads lanes that must be active).
// Wait states: s_nop 1 in front (VALU write of a VGPR -> DPP read: 2) and behind; VCC is written by v_cmp (VALU) and read by
// v_cndmask as its mask: 2 wait states on gfx940 / gfx950 (see row_scan_step64) -- s_nop 1 behind each v_cmp.  VCC is clobbered.
#define NDZIP_SWAPSEL(n) "v_cndmask_b32_dpp %[lo" #n "], %[a" #n "], %[b" #n "], vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
#define NDZIP_SWAPSEL_HI(n) "v_cndmask_b32_dpp %[hi" #n "], %[b" #n "], %[a" #n "], vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
NDZIP_DEV void pair_exchange_select4(uint32_t odd_flag, const uint32_t (&a)[4], const uint32_t (&b)[4], uint32_t (&lo)[4], uint32_t (&hi)[4]) {
#ifdef NDZIP_NO_EXEC_ASM
#pragma unroll
    for (int j = 0; j < 4; ++j) {  // (0xb1 = quad_perm:[1,0,3,2])
        const uint32_t other_a = static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(a[j]), 0xb1, 0xf, 0xf, true));
        const uint32_t other_b = static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(b[j]), 0xb1, 0xf, 0xf, true));
        lo[j] = odd_flag ? b[j] : other_a;
        hi[j] = odd_flag ? other_b : a[j];
    }
    return;
#endif
    asm volatile("s_nop 1\n\t"
                 "v_cmp_ne_u32_e32 vcc, 0, %[odd]\n\ts_nop 1\n\t"  // vcc = odd lanes: lo = vcc ? b : swap(a)
                 NDZIP_SWAPSEL(0) NDZIP_SWAPSEL(1) NDZIP_SWAPSEL(2) NDZIP_SWAPSEL(3)
                 "v_cmp_eq_u32_e32 vcc, 0, %[odd]\n\ts_nop 1\n\t"  // vcc = even lanes: hi = vcc ? a : swap(b)
                 NDZIP_SWAPSEL_HI(0) NDZIP_SWAPSEL_HI(1) NDZIP_SWAPSEL_HI(2) NDZIP_SWAPSEL_HI(3)
                 "s_nop 1"
                 : [lo0] "=&v"(lo[0]), [lo1] "=&v"(lo[1]), [lo2] "=&v"(lo[2]), [lo3] "=&v"(lo[3]), [hi0] "=&v"(hi[0]), [hi1] "=&v"(hi[1]),
                   [hi2] "=&v"(hi[2]), [hi3] "=&v"(hi[3])
                 : [odd] "v"(odd_flag), [a0] "v"(a[0]), [a1] "v"(a[1]), [a2] "v"(a[2]), [a3] "v"(a[3]), [b0] "v"(b[0]), [b1] "v"(b[1]),
                   [b2] "v"(b[2]), [b3] "v"(b[3])
                 : "vcc");
}
#undef NDZIP_SWAPSEL_HI
#undef NDZIP_SWAPSEL

// The stores of lds_append_nonzero are invisible to the compiler's s_waitcnt bookkeeping (inline asm is opaque to it): the
// workgroup barrier behind which other wavefronts read the compacted run must be preceded by this explicit wait.  (In the builds
// looked at the compiler had an lgkmcnt(0) of its own in front of that barrier -- for the chunk head it stores itself -- but
// nothing obliges it to.)
NDZIP_DEV void lds_append_complete() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

}  // namespace ndzip_hip
