// ndzip_amd/csrc/codec_kernels.hpp -- gfx950 device code for the ndzip block encode/decode hot path.
//
// What is computed (bit-exact with the reference):
//   encode: rotl1 -> integer Lorenzo differences along every axis -> complement_negative -> per chunk of B
//           values: head = OR, BxB bit-plane transpose, non-zero planes appended      (reference CPU:
//           src/ndzip/cpu_codec.inl:74-85,325-332,541-559; reference CUDA: src/ndzip/cuda_codec.inl:30-126,185-275)
//   decode: the inverse (cpu_codec.inl:561-578,335-341,87-98; cuda_codec.inl:278-365,129-183,58-65)
//
// How it is mapped on CDNA4 (none of this mirrors the reference's warp-32 kernels):
//   * 128 work-items (2 wavefronts) own one hypercube; work-item t owns the 32 consecutive cube-local values
//     [32t, 32t+32).  For f32 that is exactly one chunk, for f64 half a chunk (lanes 2m, 2m+1 pair up).
//   * The cube is staged once in LDS with a 16-byte pad after every 32 values, which makes every
//     "work-item reads/writes its own 32 values with ds_*_b128" access bank-conflict free on the 64-bank LDS.
//   * The forward Lorenzo transform is evaluated as one fused stencil straight out of that staging buffer
//     (out-of-cube neighbours are read from a zero block), so there is a single barrier instead of one per axis.
//   * The bit-plane transpose is a 5-stage in-register block-swap network per work-item (2x v_perm_b32 stages,
//     3x shift+v_bfi stages): ~8 VALU ops per value instead of ~64 for a ballot per plane.
//   * Chunk offsets inside a hypercube are a 128-wide scan done with wave shuffles; only ONE length per
//     tile takes part in the device-wide scan, which is a decoupled look-back fused into the same kernel
//     (traffic N + C instead of the reference's N + 3C).
//   * The encoded hypercube is compacted in LDS and leaves as one contiguous, coalesced run.
#pragma once

#include <hip/hip_runtime.h>

#include "codec_common.hpp"
#include "gfx950_lds.hpp"

namespace ndzip_hip {

// ---------------------------------------------------------------------------------------------------------
// small bit helpers (common.hh:436-449)
// ---------------------------------------------------------------------------------------------------------

NDZIP_DEV uint32_t rotl1(uint32_t v) { return __builtin_amdgcn_alignbit(v, v, 31); }
NDZIP_DEV uint32_t rotr1(uint32_t v) { return __builtin_amdgcn_alignbit(v, v, 1); }
// (64 bits: two v_alignbit_b32 over the register pair instead of a 64-bit shift, a 32-bit shift and an OR)
NDZIP_DEV uint64_t rotl1(uint64_t v) {
    const uint32_t hi = static_cast<uint32_t>(v >> 32), lo = static_cast<uint32_t>(v);
    return (static_cast<uint64_t>(__builtin_amdgcn_alignbit(hi, lo, 31)) << 32) | __builtin_amdgcn_alignbit(lo, hi, 31);
}
NDZIP_DEV uint64_t rotr1(uint64_t v) {
    const uint32_t hi = static_cast<uint32_t>(v >> 32), lo = static_cast<uint32_t>(v);
    return (static_cast<uint64_t>(__builtin_amdgcn_alignbit(lo, hi, 1)) << 32) | __builtin_amdgcn_alignbit(hi, lo, 1);
}

// v >> (B-1) ? v ^ (~0 >> 1) : v   ==   v ^ (sign_mask & (~0 >> 1)),  sign_mask = sint(v) >> (B-1)
// The mask goes through opaque_vgpr so that the optimiser cannot canonicalise `ashr ; and` into `ashr ; lshr ; xor` (three
// instructions per value; five for 64 bits, two of them on a 64-bit shift): as written it selects v_ashrrev_i32 +
// v_bitop3_b32 (x ^ (m & c)) -- and for 64 bits one v_ashrrev_i32 of the high half, v_xor of the low half, v_bitop3 of the
// high half.  32 (f32) / 16 (f64) values per work-item and hypercube.
NDZIP_DEV uint32_t complement_negative(uint32_t v) {
    const uint32_t m = static_cast<uint32_t>(opaque_vgpr(static_cast<int32_t>(v) >> 31));
    return v ^ (m & 0x7fffffffu);
}
NDZIP_DEV uint64_t complement_negative(uint64_t v) {
    const uint32_t hi = static_cast<uint32_t>(v >> 32), lo = static_cast<uint32_t>(v);
    const uint32_t m = static_cast<uint32_t>(opaque_vgpr(static_cast<int32_t>(hi) >> 31));
    return (static_cast<uint64_t>(hi ^ (m & 0x7fffffffu)) << 32) | (lo ^ m);
}

NDZIP_DEV int popcount_w(uint32_t v) { return __builtin_popcount(v); }
NDZIP_DEV int popcount_w(uint64_t v) { return __builtin_popcountll(v); }

// ---------------------------------------------------------------------------------------------------------
// 32x32 bit transpose, "mirrored" indexing of the format: out[i] bit (31-j) = in[j] bit (31-i)
// (cpu_codec.inl:355-363).  Viewing word r as matrix row r with the MSB as column 0 this is the plain
// matrix transpose; stage s swaps the off-diagonal s x s blocks.  The 64x64 transpose of f64 chunks is
// four of these (see encode/decode below).
// ---------------------------------------------------------------------------------------------------------

template<int S, uint32_t M, int N = 32>
NDZIP_DEV void swap_stage(uint32_t (&x)[N]) {
#pragma unroll
    for (int r = 0; r < N; ++r) {
        if ((r & S) == 0) {
            const uint32_t a = x[r], b = x[r + S];
            x[r] = (a & ~M) | ((b >> S) & M);      // v_lshrrev + v_bfi
            x[r + S] = ((a << S) & ~M) | (b & M);  // v_lshlrev + v_bfi
        }
    }
}

NDZIP_DEV void transpose32(uint32_t (&x)[32]) {
    // stages 16 and 8 move whole bytes: one v_perm_b32 per output word.
    // __builtin_amdgcn_perm(hi, lo, sel): selector byte 0-3 picks lo.byte[0-3], 4-7 picks hi.byte[0-3].
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const uint32_t a = x[r], b = x[r + 16];
        x[r] = __builtin_amdgcn_perm(a, b, 0x07060302u);       // (a & 0xffff0000) | (b >> 16)
        x[r + 16] = __builtin_amdgcn_perm(a, b, 0x05040100u);  // (a << 16) | (b & 0x0000ffff)
    }
#pragma unroll
    for (int r = 0; r < 32; ++r) {
        if ((r & 8) == 0) {
            const uint32_t a = x[r], b = x[r + 8];
            x[r] = __builtin_amdgcn_perm(a, b, 0x07030501u);      // (a & 0xff00ff00) | ((b >> 8) & 0x00ff00ff)
            x[r + 8] = __builtin_amdgcn_perm(a, b, 0x06020400u);  // ((a << 8) & 0xff00ff00) | (b & 0x00ff00ff)
        }
    }
    swap_stage<4, 0x0f0f0f0fu>(x);
    swap_stage<2, 0x33333333u>(x);
    swap_stage<1, 0x55555555u>(x);
}

// Portable shift/mask form of the same network; the unit test checks both against the oracle.
NDZIP_DEV void transpose32_generic(uint32_t (&x)[32]) {
    swap_stage<16, 0x0000ffffu>(x);
    swap_stage<8, 0x00ff00ffu>(x);
    swap_stage<4, 0x0f0f0f0fu>(x);
    swap_stage<2, 0x33333333u>(x);
    swap_stage<1, 0x55555555u>(x);
}

// ---------------------------------------------------------------------------------------------------------
// LDS staging layout: value k lives at byte k*sizeof(W) + (k/32)*16.
// ---------------------------------------------------------------------------------------------------------

template<typename W>
struct lds_layout {
    static constexpr uint32_t chunk_bytes = 32 * sizeof(W) + 16;
    static constexpr uint32_t cube_bytes = threads_per_hc * chunk_bytes;  // 18432 (f32) / 34816 (f64)
    // Region holding the zero block that out-of-cube stencil neighbours read.  The block itself sits at
    // zero_offset<Dims>() inside the region: the 16-byte slot (mod 256) that none of the in-cube lanes of the same
    // 16-lane group touches in any of the stencil's neighbour reads (tools/ldsbench3.hip scans all 16 slots:
    // f32 3D {3, 11}, f64 3D {7, 15} for rows y-1 / z-1,y-1; f32 2D {7, 14}, f64 2D {14, 15}).  With the block
    // in slot 0 every neighbour read of the 3D stencil paid a 2-way conflict in each lane group.
    static constexpr uint32_t zero_bytes = 32 * sizeof(W) + 256;
    template<int Dims>
    NDZIP_DEV static constexpr uint32_t zero_offset() {
        return sizeof(W) == 8 ? 15 * 16 : Dims == 3 ? 3 * 16 : Dims == 2 ? 7 * 16 : 0;
    }
    NDZIP_DEV static constexpr uint32_t off(uint32_t k) { return k * static_cast<uint32_t>(sizeof(W)) + (k >> 5) * 16u; }
};

NDZIP_DEV void lds_write16(char *p, vec16 v) { *reinterpret_cast<vec16 *>(p) = v; }

// Where the ENCODED RUN of a hypercube lives inside its (128-byte aligned) LDS region.  The compaction (encoder) and the
// gather (decoder) touch it one 4-byte word per lane at a lane stride of "plane words the chunk keeps" -- for 64-bit profiles
// 2 x kept planes dwords, an even number, and when most chunks of a wavefront keep the same count and that stride is a multiple
// of 32 dwords (48 of 64 planes on the benchmark's 3D f64 grid) all 16 chunks of a lane group meet in the same banks:
// tools/lds_profile.py prices that gather at 11x and the compaction at 9.5x their conflict-free cycles.  So 64-bit runs are
// stored with the 16-byte slot s of every 128-byte block b at slot s ^ (b & 7): lane strides of 32 k dwords then spread over 8
// slots, and the 16-byte staging stores and copy-out reads still move whole slots.  32-bit runs stay linear: their stride is
// "planes kept", odd as often as even, and a swizzle makes the benchmark's 19-plane chunks slightly worse (2.7x -> 3.3x,
// simulated over the oracle's streams).
// The swizzle works on LDS byte ADDRESSES (gfx950_lds.hpp: lds_address / lds_pointer), so a region only has to start on a
// 128-byte boundary of the LDS, and it costs a shift and one three-input bit operation per access.
template<typename W>
struct run_layout {
    static constexpr bool swizzled = sizeof(W) == 8;
    // where the byte with the linear LDS address `a` is kept
    NDZIP_DEV static constexpr uint32_t at(uint32_t a) {
        if constexpr (swizzled) {
            return a ^ ((a >> 3) & 0x70u);
        } else {
            return a;
        }
    }
    // the same for a pointer into the region
    template<typename P>
    NDZIP_DEV static P *ptr(P *linear) {
        if constexpr (swizzled) {
            return reinterpret_cast<P *>(lds_pointer(at(lds_address(linear))));
        } else {
            return linear;
        }
    }
};

// 16 / 8 bytes per lane from / to GLOBAL memory.  Aligned = the address is a multiple of the access size; otherwise only
// of the element size (a row of an array whose extent is not a multiple of 4 elements starts anywhere): gfx950 runs in
// unaligned access mode, so this is still ONE global_load/store_dwordx4 per lane -- it may touch one cache line more per
// row, nothing like the 4-byte-per-lane accesses of a scalar path.
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
struct __attribute__((packed, aligned(4))) unaligned16 {
    u32x4_t v;
};
struct __attribute__((packed, aligned(4))) unaligned8 {
    u32x2_t v;
};
// Array input is read exactly once per launch: the aligned path uses the read-once (nt) policy (gfx950_lds.hpp).  Once = false: the caller reads only part of every cache
// line and another workgroup the rest (the 64-byte rows of an unpaired 3D f32 hypercube), so the line should stay cacheable.
template<bool Aligned, bool Once = true>
NDZIP_DEV vec16 global_load16(const void *p) {
    if constexpr (Aligned && !Once) {
        return *reinterpret_cast<const vec16 *>(p);
    } else if constexpr (Aligned) {
        return global_load16_once(p);
    } else {
        const u32x4_t q = reinterpret_cast<const unaligned16 *>(p)->v;
        vec16 v;
        v.w[0] = q.x;
        v.w[1] = q.y;
        v.w[2] = q.z;
        v.w[3] = q.w;
        return v;
    }
}
template<bool Aligned>
NDZIP_DEV void global_store16(void *p, const vec16 &v) {
    if constexpr (Aligned) {
        *reinterpret_cast<vec16 *>(p) = v;
    } else {
        unaligned16 u;
        u.v = u32x4_t{v.w[0], v.w[1], v.w[2], v.w[3]};
        *reinterpret_cast<unaligned16 *>(p) = u;
    }
}
template<bool Aligned>
NDZIP_DEV void global_store8(void *p, uint32_t a, uint32_t b) {
    if constexpr (Aligned) {
        *reinterpret_cast<uint2 *>(p) = make_uint2(a, b);
    } else {
        unaligned8 u;
        u.v = u32x2_t{a, b};
        *reinterpret_cast<unaligned8 *>(p) = u;
    }
}

template<typename W>
NDZIP_DEV W lds_read(const char *base, uint32_t byte_off) {
    return *reinterpret_cast<const W *>(base + byte_off);
}

// N consecutive values from a 16-byte aligned LDS address that does not cross a chunk pad (N * sizeof(W) % 16 == 0)
template<typename W, int N>
NDZIP_DEV void read_run(const char *p, W (&dst)[N]) {
    constexpr int per = 16 / sizeof(W);  // values per 16-byte read
    static_assert(N % per == 0, "whole 16-byte vectors only");
#pragma unroll
    for (int i = 0; i < N / per; ++i) {
        const vec16 v = lds_read16(p + 16 * i);
        if constexpr (sizeof(W) == 4) {
#pragma unroll
            for (int j = 0; j < 4; ++j) dst[4 * i + j] = v.w[j];
        } else {
            dst[2 * i] = static_cast<uint64_t>(v.w[0]) | (static_cast<uint64_t>(v.w[1]) << 32);
            dst[2 * i + 1] = static_cast<uint64_t>(v.w[2]) | (static_cast<uint64_t>(v.w[3]) << 32);
        }
    }
}

template<typename W>
NDZIP_DEV void read_run16(const char *p, W (&dst)[16]) {
    read_run<W, 16>(p, dst);
}

template<typename W>
NDZIP_DEV void write_run16(char *p, const W (&src)[16]) {
    constexpr int per = 16 / sizeof(W);
#pragma unroll
    for (int i = 0; i < 16 / per; ++i) {
        vec16 v;
        if constexpr (sizeof(W) == 4) {
#pragma unroll
            for (int j = 0; j < 4; ++j) v.w[j] = src[4 * i + j];
        } else {
            v.w[0] = static_cast<uint32_t>(src[2 * i]);
            v.w[1] = static_cast<uint32_t>(src[2 * i] >> 32);
            v.w[2] = static_cast<uint32_t>(src[2 * i + 1]);
            v.w[3] = static_cast<uint32_t>(src[2 * i + 1] >> 32);
        }
        lds_write16(p + 16 * i, v);
    }
}

// ---------------------------------------------------------------------------------------------------------
// hypercube <-> global addressing (common.hh:538-579)
// ---------------------------------------------------------------------------------------------------------

// n / d and n % d with the precomputed magic = floor(2^32 / d): the estimate mulhi(n, magic) is exact or one
// too small, so a single correction step suffices (no v_rcp / integer-division expansion per tile).
NDZIP_DEV void fast_divmod(uint32_t n, uint32_t d, uint32_t magic, uint32_t &q, uint32_t &r) {
    q = __umulhi(n, magic);
    r = n - q * d;
    if (r >= d) {
        ++q;
        r -= d;
    }
}

template<int Dims>
NDZIP_DEV uint64_t hc_origin(const grid_geom &gg, uint32_t hc) {
    constexpr uint32_t side = side_of<Dims>::value;
    uint64_t off = 0;
#pragma unroll
    for (int nd = 0; nd < Dims; ++nd) {
        const int d = Dims - 1 - nd;
        uint32_t c;
        if (d == 0) {
            c = hc;  // slowest dimension: nothing left to divide off
        } else {
            uint32_t q;
            fast_divmod(hc, gg.g[d], gg.g_magic[d], q, c);
            hc = q;
        }
        off += static_cast<uint64_t>(c) * side * gg.stride[d];
    }
    return off;
}

// element offset of cube-local index k relative to the hypercube origin
template<int Dims>
NDZIP_DEV uint64_t local_offset(const grid_geom &gg, uint32_t k) {
    if constexpr (Dims == 1) {
        return k;
    } else if constexpr (Dims == 2) {
        return static_cast<uint64_t>(k >> 6) * gg.stride[0] + (k & 63u);
    } else {
        return static_cast<uint64_t>(k >> 8) * gg.stride[0] + static_cast<uint64_t>((k >> 4) & 15u) * gg.stride[1]
                + (k & 15u);
    }
}

// local_offset<Dims>(gg, k0) of a work-item's first value k0 = t * (values per work-item access), as a wave-uniform element
// offset (3D: the z-plane, which 64 consecutive work-items share whenever they cover at most 256 consecutive values) plus a
// per-lane BYTE offset inside that plane / within the first rows.  The byte offset fits 32 bits for every legal array: an array
// has at most 2^32 - 1 elements and at least `side` rows / planes, so 16 rows of a 3D plane are < 2^24 * 16 elements and 8 rows
// of a 2D array < 2^26 * 8.
template<int Dims, typename W>
NDZIP_DEV void split_local_offset(const grid_geom &gg, uint32_t k0, uint64_t &uniform_elems, uint32_t &lane_bytes) {
    if constexpr (Dims == 1) {
        uniform_elems = 0;
        lane_bytes = k0 * static_cast<uint32_t>(sizeof(W));
    } else if constexpr (Dims == 2) {
        uniform_elems = 0;
        lane_bytes = ((k0 >> 6) * static_cast<uint32_t>(gg.stride[0]) + (k0 & 63u)) * static_cast<uint32_t>(sizeof(W));
    } else {
        uniform_elems = static_cast<uint64_t>(static_cast<uint32_t>(wave_uniform(static_cast<int>(k0 >> 8)))) * gg.stride[0];
        lane_bytes = (((k0 >> 4) & 15u) * static_cast<uint32_t>(gg.stride[1]) + (k0 & 15u)) * static_cast<uint32_t>(sizeof(W));
    }
}

// ---------------------------------------------------------------------------------------------------------
// wave / group scans
// ---------------------------------------------------------------------------------------------------------

// Value of lane (l - D) for the lanes of an 8-lane group (l % 8 >= D), 0 for the others, as a DPP row shift instead of
// ds_bpermute + select: row_shr:D moves within 16-lane rows (bound_ctrl zero-fills at the row start); the lanes of the
// second group of a row that would read across the group boundary are cleared with `keep` (all ones iff l % 8 >= D) or,
// for D == 4, by the bank mask alone.
template<int D>
NDZIP_DEV uint32_t group8_shift_up(uint32_t v, uint32_t keep) {
    static_assert(D == 1 || D == 2 || D == 4);
    if constexpr (D == 4) {
        return static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x110 + D, 0xf, 0xa, false));
    } else {
        return static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x110 + D, 0xf, 0xf, true)) & keep;
    }
}
template<int D>
NDZIP_DEV uint64_t group8_shift_up(uint64_t v, uint32_t keep) {
    const uint32_t lo = group8_shift_up<D>(static_cast<uint32_t>(v), keep);
    const uint32_t hi = group8_shift_up<D>(static_cast<uint32_t>(v >> 32), keep);
    return (static_cast<uint64_t>(hi) << 32) | lo;
}

// Inclusive prefix sum over the 64 lanes with DPP: four row shifts (lanes shifted in from outside a 16-lane row read as 0)
// scan each row, row_bcast:15 adds the total of row 0 / 2 to row 1 / 3, row_bcast:31 the total of rows 0-1 to rows 2-3 --
// six VALU additions with no LDS round trip.  (__shfl_up compiles to ds_bpermute_b32 plus index arithmetic and a select
// per step: six DEPENDENT LDS-crossbar round trips, ~700 cycles, on the path to the aggregate publish.)
template<int Ctrl, int RowMask>
NDZIP_DEV uint32_t dpp_or_zero(uint32_t v) {
    return static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), Ctrl, RowMask, 0xf, false));
}
NDZIP_DEV uint32_t wave_inclusive_scan(uint32_t v, int /* lane */) {
    v += dpp_or_zero<0x111, 0xf>(v);  // row_shr:1
    v += dpp_or_zero<0x112, 0xf>(v);  // row_shr:2
    v += dpp_or_zero<0x114, 0xf>(v);  // row_shr:4
    v += dpp_or_zero<0x118, 0xf>(v);  // row_shr:8
    v += dpp_or_zero<0x142, 0xa>(v);  // row_bcast:15 into rows 1 and 3
    v += dpp_or_zero<0x143, 0xc>(v);  // row_bcast:31 into rows 2 and 3
    return v;
}
// sum over the 64 lanes, wave-uniform (the scan's last lane)
NDZIP_DEV uint32_t wave_sum(uint32_t v) {
    return static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(wave_inclusive_scan(v, 0)), 63));
}

// the same for the value type's word: 32 bits = the scan above; 64 bits = its twelve-instruction counterpart with the carry taken along
// (gfx950_lds.hpp: wave_inclusive_scan64) -- rounds 1-3 ran six __shfl_up steps here (ds_bpermute_b32 pairs, compare, selects)
NDZIP_DEV uint32_t wave_inclusive_scan_w(uint32_t v, int lane) { return wave_inclusive_scan(v, lane); }
NDZIP_DEV uint64_t wave_inclusive_scan_w(uint64_t v, int /* lane */) {
    uint32_t lo = static_cast<uint32_t>(v), hi = static_cast<uint32_t>(v >> 32);
    wave_inclusive_scan64(lo, hi);
    return (static_cast<uint64_t>(hi) << 32) | lo;
}

// ---------------------------------------------------------------------------------------------------------
// ENCODE one hypercube with 128 work-items.
//   in        global array (as words)            origin   element offset of the hypercube
//   cube      this hypercube's LDS staging region (lds_layout<W>::cube_bytes), reused for the encoded run
//   zero      LDS zero block (lds_layout<W>::zero_bytes, 16-byte aligned)
//   xchg      2 x uint32 LDS scratch for this hypercube
// Returns the encoded length in words (uniform across the 128 work-items); the words sit at cube[0 .. L).
// Contains __syncthreads(): every work-item of the workgroup must call it (`active` = false for padding
// groups, which still take part in the barriers).
// ---------------------------------------------------------------------------------------------------------

// raw input of one work-item, held in registers between the global load and the LDS staging (this is what the
// persistent compress kernel prefetches for the NEXT tile while the current one is being written out)
template<typename W, bool Aligned>
struct input_regs {
    static constexpr int VE = 16 / sizeof(W);                  // values per 16-byte vector
    static constexpr int NV = hc_size / VE / threads_per_hc;   // 8 (f32) / 16 (f64) vectors per work-item
    vec16 v[NV];
};

// phase 0a: issue the coalesced global loads of hypercube `origin` (nothing waits here)
// Part 0 = the first `Split` vectors (values) of the work-item, part 1 = the rest, part -1 = everything: the deferred
// write-out kernel issues part 0 before the stencil (a full iteration ahead) and part 1 after it, which caps the
// registers that are live across the stencil.
template<typename T, int Dims, bool Aligned, int Part = -1, int Split = 0>
NDZIP_DEV void load_hypercube_regs(const typename profile<T, Dims>::word *__restrict__ in, const grid_geom &gg,
        uint64_t origin, int t, input_regs<typename profile<T, Dims>::word, Aligned> &regs) {
    using W = typename profile<T, Dims>::word;
    using R = input_regs<W, Aligned>;
    constexpr int first = Part == 1 ? Split : 0;
    constexpr int last = Part == 0 ? Split : R::NV;
    // vector i of work-item t covers cube-local values k_i = (i*128 + t) * VE; 128*VE values are a whole number of
    // rows / planes, so the global offset is affine in i: one per-lane base plus a uniform step.  A vector never
    // crosses a hypercube row (VE divides the side length), so the same path serves unaligned rows.
    // The address is split as in load_pair_regs: wave-uniform 64-bit base (hypercube origin, the z-plane this wavefront reads,
    // i steps) + 32-bit per-lane byte offset inside a plane.
    uint64_t plane;
    uint32_t lane_bytes;
    split_local_offset<Dims, W>(gg, static_cast<uint32_t>(t) * R::VE, plane, lane_bytes);
    const char *base = reinterpret_cast<const char *>(in + origin + plane);
    const uint64_t step = local_offset<Dims>(gg, threads_per_hc * R::VE) * sizeof(W);
    constexpr bool whole_lines = !(Dims == 3 && sizeof(W) == 4);  // (3D f32 rows are 64 bytes: half a line each)
    const uint32_t off = lane_offset_here(lane_bytes);
    // (a running scalar pointer: one 64-bit step in SGPRs instead of seven precomputed multiples of it)
    const char *p = scalar_pointer(base + first * step);
#pragma unroll
    for (int i = first; i < last; ++i) {
        regs.v[i] = global_load16<Aligned, whole_lines>(p + off);
        if (i + 1 < last) p = scalar_pointer(p + step);
    }
}

// phase 0b: rotl1 and store to the padded LDS staging layout
template<typename W, bool Aligned>
NDZIP_DEV void stage_hypercube_regs(const input_regs<W, Aligned> &regs, char *cube, int t) {
    using L = lds_layout<W>;
    using R = input_regs<W, Aligned>;
    {
        // vector i sits at value (i*128 + t) * VE; 128 * VE values are whole padded chunks, so the LDS address is one
        // per-lane base plus a compile-time step (an immediate offset of ds_write_b128, not eight address registers)
        char *base = cube + L::off(static_cast<uint32_t>(t) * R::VE);
        constexpr uint32_t step = L::off(threads_per_hc * R::VE);
#pragma unroll
        for (int i = 0; i < R::NV; ++i) {
            vec16 r;
            if constexpr (sizeof(W) == 4) {
#pragma unroll
                for (int j = 0; j < 4; ++j) r.w[j] = rotl1(regs.v[i].w[j]);
            } else {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const uint64_t x = rotl1(static_cast<uint64_t>(regs.v[i].w[2 * j])
                            | (static_cast<uint64_t>(regs.v[i].w[2 * j + 1]) << 32));
                    r.w[2 * j] = static_cast<uint32_t>(x);
                    r.w[2 * j + 1] = static_cast<uint32_t>(x >> 32);
                }
            }
            lds_write16(base + i * step, r);
        }
    }
}

// ---- paired variant (3D, 32-bit words): a tile of two hypercubes that are neighbours along x is 256 rows of 128
// contiguous bytes.  256 work-items fetch it as whole 128-byte rows -- vector i of work-item `tid` is 16-byte piece
// tid % 8 of row i*32 + tid/8 (pieces 0-3 belong to the first cube, 4-7 to the second) -- so every wave-instruction
// covers 8 full cache lines instead of 16 half lines (bare load loop: 0.098-0.101 ms vs 0.108-0.115 ms for 512^3 f32,
// tools/membench.hip).  Offsets are affine in i: two z-planes per step in global memory, 2304 bytes in LDS.
// `cube_stride` (bytes between the two cubes' staging regions) must be = 64 mod 128: a ds_write_b128 is served in groups of 8
// consecutive lanes over 32 banks (MI355X guide, LDS table) = the 8 pieces of one row, 64 bytes for each cube -- which land on
// the two halves of the bank space only if the cubes sit an odd multiple of 64 bytes apart.  (Round 1 had 128 mod 256, designed
// for 64 banks: every staging write was a 2-way conflict, 128 of the 269 conflict cycles per hypercube of the 3D f32 kernel --
// tools/lds_profile.py; its total of 33.9 % conflict cycles matches the 35 % the round-1 PMC run measured.)
template<int Part = -1, int Split = 0>
NDZIP_DEV void load_pair_regs(const uint32_t *__restrict__ in, const grid_geom &gg, uint64_t pair_origin, int tid,
        input_regs<uint32_t, true> &regs) {
    using R = input_regs<uint32_t, true>;
    constexpr int first = Part == 1 ? Split : 0;
    constexpr int last = Part == 0 ? Split : R::NV;
    // address = (wave-uniform 64-bit base: the tile, this wave pair's z-plane, the step) + (32-bit per-lane byte offset inside one
    // z-plane: loop-invariant) -- so that the base arithmetic runs on the scalar unit and the load takes the SGPR-base + VGPR-offset
    // form instead of a 64-bit per-lane pointer advanced by 64-bit VALU additions (a z-plane of a legal array is < 2^30 bytes:
    // at most 2^32 - 1 elements over at least 16 planes)
    const uint32_t r0 = static_cast<uint32_t>(tid) >> 3, piece = static_cast<uint32_t>(tid) & 7u;
    const uint32_t zw = static_cast<uint32_t>(wave_uniform(static_cast<int>(r0 >> 4)));
    const uint32_t lane_bytes = ((r0 & 15u) * static_cast<uint32_t>(gg.stride[1]) + piece * 4u) * 4u;
    const char *base = reinterpret_cast<const char *>(in + pair_origin + static_cast<uint64_t>(zw) * gg.stride[0]);
    const uint64_t step = 2 * gg.stride[0] * sizeof(uint32_t);
    const uint32_t off = lane_offset_here(lane_bytes);
    const char *p = scalar_pointer(base + first * step);  // (a running scalar pointer: see load_hypercube_regs)
#pragma unroll
    for (int i = first; i < last; ++i) {
        regs.v[i] = global_load16<true>(p + off);
        if (i + 1 < last) p = scalar_pointer(p + step);
    }
}

NDZIP_DEV void stage_pair_regs(const input_regs<uint32_t, true> &regs, char *cubes, uint32_t cube_stride, int tid) {
    using L = lds_layout<uint32_t>;
    using R = input_regs<uint32_t, true>;
    const uint32_t r0 = static_cast<uint32_t>(tid) >> 3, piece = static_cast<uint32_t>(tid) & 7u;
    char *base = cubes + (piece >> 2) * cube_stride + L::off(r0 * 16u + (piece & 3u) * 4u);
    constexpr uint32_t step = L::off(32 * 16);  // 32 rows
#pragma unroll
    for (int i = 0; i < R::NV; ++i) {
        vec16 r;
#pragma unroll
        for (int j = 0; j < 4; ++j) r.w[j] = rotl1(regs.v[i].w[j]);
        lds_write16(base + i * step, r);
    }
}

// phase 1: fused Lorenzo stencil out of the staged cube + complement_negative -> residuals r[32] of work-item t.
// No barrier inside: the caller orders it after the staging writes and before the cube is overwritten.
template<typename T, int Dims>
NDZIP_DEV void stencil_residuals(const char *cube, const char *zero, int t, typename profile<T, Dims>::word (&r)[vals_per_thread]) {
    using P = profile<T, Dims>;
    using W = typename P::word;
    using L = lds_layout<W>;
    // ---- phase 1: fused Lorenzo stencil out of LDS -> residuals r[32] in registers ----------------------
    const uint32_t k0 = static_cast<uint32_t>(t) * 32u;
    const char *own = cube + L::off(k0);
    if constexpr (Dims == 1) {
        W o[32];
        read_run16<W>(own, *reinterpret_cast<W(*)[16]>(&o[0]));
        read_run16<W>(own + 16 * sizeof(W), *reinterpret_cast<W(*)[16]>(&o[16]));
        // predecessor of the first value: last value of chunk t-1 (0 for the first value of the cube)
        const W prev = lds_read<W>(t > 0 ? cube + L::off(k0 - 1) : zero, 0);
#pragma unroll
        for (int j = 31; j >= 1; --j) r[j] = o[j] - o[j - 1];
        r[0] = o[0] - prev;
    } else if constexpr (Dims == 2) {
        // chunk = half a row: y = t / 2, h = t % 2; folded in two runs of 16 values (two runs live at a time)
        const int y = t >> 1, h = t & 1;
        const char *up = y > 0 ? cube + L::off(k0 - 64) : zero;
        const W ol = lds_read<W>(h ? cube + L::off(k0 - 1) : zero, 0);
        const W ul = lds_read<W>((h && y > 0) ? cube + L::off(k0 - 65) : zero, 0);
        W left = ol - ul;  // y difference at x - 1
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            W o[16], u[16];
            read_run<W, 16>(own + q * 16 * sizeof(W), o);
            read_run<W, 16>(up + q * 16 * sizeof(W), u);
#pragma unroll
            for (int j = 0; j < 16; ++j) o[j] -= u[j];  // y difference
#pragma unroll
            for (int j = 15; j >= 1; --j) r[16 * q + j] = o[j] - o[j - 1];  // x difference
            r[16 * q] = o[0] - left;
            left = o[15];
            __builtin_amdgcn_sched_barrier(0);
        }
    } else {
        // chunk = rows (z, y0) and (z, y0 + 1): z = t / 8, y0 = 2 * (t % 8)
        const int z = t >> 3, yp = t & 7;
        const char *row_p = yp > 0 ? cube + L::off(k0 - 16) : zero;                   // (z, y0-1)
        const char *row_a1 = z > 0 ? cube + L::off(k0 - 256) : zero;                  // (z-1, y0)
        const char *row_b1 = z > 0 ? row_a1 + 16 * sizeof(W) : zero;                  // (z-1, y0+1)
        const char *row_p1 = (z > 0 && yp > 0) ? cube + L::off(k0 - 256 - 16) : zero; // (z-1, y0-1)
        // The two rows are folded in halves of 8 values with scheduling barriers in between: at most four half-rows
        // are live at a time.  Left alone the scheduler issues all 24 16-byte reads up front; as whole rows the fold
        // needed ~30 more VGPRs at its peak, which (next to the previous tile's 32 plane words) spilled -- and a scratch
        // reload waits vmcnt(0), i.e. for every prefetch load in flight.
        constexpr int H = 8;
        W carry_a = 0, carry_b = 0;  // last value of the previous half (x - 1 of this half's first value)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const uint32_t o = h * H * sizeof(W);
            W a[H], b[H];
            read_run<W, H>(own + o, a);
            read_run<W, H>(own + 16 * sizeof(W) + o, b);
#pragma unroll
            for (int j = 0; j < H; ++j) b[j] -= a[j];  // row y0+1 minus row y0
            __builtin_amdgcn_sched_barrier(0);
            {
                W p[H];
                read_run<W, H>(row_p + o, p);
#pragma unroll
                for (int j = 0; j < H; ++j) a[j] -= p[j];  // row y0 minus row y0-1
            }
            __builtin_amdgcn_sched_barrier(0);
            {
                W a1[H];
                read_run<W, H>(row_a1 + o, a1);
                {
                    W b1[H];
                    read_run<W, H>(row_b1 + o, b1);
#pragma unroll
                    for (int j = 0; j < H; ++j) b[j] -= b1[j] - a1[j];
                }
                __builtin_amdgcn_sched_barrier(0);
                {
                    W p1[H];
                    read_run<W, H>(row_p1 + o, p1);
#pragma unroll
                    for (int j = 0; j < H; ++j) a[j] -= a1[j] - p1[j];
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = H - 1; j >= 1; --j) {
                r[h * H + j] = a[j] - a[j - 1];
                r[16 + h * H + j] = b[j] - b[j - 1];
            }
            r[h * H] = a[0] - carry_a;
            r[16 + h * H] = b[0] - carry_b;
            carry_a = a[H - 1];
            carry_b = b[H - 1];
        }
    }
#pragma unroll
    for (int j = 0; j < 32; ++j) r[j] = complement_negative(r[j]);
}

// phases 0+1 for one hypercube (stage tests and the simple path): load, stage, barrier, stencil, barrier
template<typename T, int Dims, bool Aligned>
NDZIP_DEV void forward_transform_hypercube(const typename profile<T, Dims>::word *__restrict__ in, const grid_geom &gg,
        uint64_t origin, bool active, char *cube, const char *zero, int t,
        typename profile<T, Dims>::word (&r)[vals_per_thread]) {
    using W = typename profile<T, Dims>::word;
    if (active) {
        input_regs<W, Aligned> regs;
        load_hypercube_regs<T, Dims, Aligned>(in, gg, origin, t, regs);
        stage_hypercube_regs<W, Aligned>(regs, cube, t);
    }
    __syncthreads();
    stencil_residuals<T, Dims>(cube, zero, t, r);
    __syncthreads();  // all stencil reads done: `cube` may now be overwritten with the encoded run
}

// ---- phase 2 of encode (f32; the f64 mapping lives in codec_kernels_wide.hpp) -----------------------------------------
// A work-item holds one chunk: 32 residuals -> head (OR), count of non-zero planes, and -- after transpose32 -- the 32
// plane words (plane 0 = MSB plane).  The kernel publishes the lengths first and writes the planes later:

// head word and plane count of the chunk held in r[32]
NDZIP_DEV uint32_t chunk_head32(const uint32_t (&r)[vals_per_thread]) {
    uint32_t head = 0;
#pragma unroll
    for (int j = 0; j < 32; ++j) head |= r[j];
    return head;
}

// Compaction of one chunk into the encoded run of its hypercube in LDS: head word at run[t], the non-zero planes from
// run[head_words + chunk_excl] on (`chunk_excl` = plane words of all earlier chunks of the hypercube).  `run_word0` = uint32
// index of run[0] inside a 16-byte aligned LDS region, for the alignment test of the dense path:
// a chunk that keeps all 32 planes at a 16-byte aligned position goes out as eight 16-byte writes -- when whole wavefronts
// are that dense (incompressible data) the word-by-word compaction writes at a lane stride of 32 words, a 32-way bank
// conflict on each of its 32 instructions (random bits: compress 0.345 -> 0.29 ms for 512^3).
NDZIP_DEV void write_planes32(uint32_t *run, uint32_t run_word0, int t, uint32_t head, uint32_t chunk_excl, const uint32_t (&planes)[32]) {
    constexpr uint32_t head_words = hc_size / 32;
    uint32_t pos = head_words + chunk_excl;
    run[t] = head;
    const bool dense = head == 0xffffffffu && ((run_word0 + pos) & 3u) == 0;
    if (dense) {
        char *dst = reinterpret_cast<char *>(run + pos);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            vec16 v;
#pragma unroll
            for (int j = 0; j < 4; ++j) v.w[j] = planes[4 * i + j];
            lds_write16(dst + 16 * i, v);
        }
    } else {
        // word-by-word compaction at a running LDS address, branch-free (gfx950_lds.hpp)
        lds_append_nonzero(lds_address(run + pos), planes);
    }
}

// ---------------------------------------------------------------------------------------------------------
// DECODE one hypercube with 128 work-items.
//   stage   LDS region holding the encoded run, words [0, L) (same region as `cube`; it is consumed before
//           the decoded values overwrite it)
//   out     global array (as words), origin = element offset of the hypercube
// ---------------------------------------------------------------------------------------------------------

// phase 1 of decode: encoded run at cube[0 .. L) -> residuals r[32] of work-item t.
// ComplementInPlaneDomain: also undo complement_negative, but BEFORE the inverse transpose: flipping the low B-1 bits of
// every negative value is "XOR every plane below the sign plane with the sign plane" -- B-1 operations on plane words
// instead of 3 per value (ashr, lshr, xor) afterwards.
// `region`: the LDS region the run was staged into, `run_off`: byte offset of the run's first word inside it (run_layout)
template<typename T, int Dims, bool ComplementInPlaneDomain = false>
NDZIP_DEV void decode_residuals(const char *region, uint32_t run_off, uint32_t *xchg, int t,
        typename profile<T, Dims>::word (&r)[vals_per_thread]) {
    using P = profile<T, Dims>;
    using R = run_layout<typename P::word>;
    constexpr int B = P::B;
    const int lane = t & 63, wave = wave_uniform(t >> 6);
    const char *cube = region + run_off;  // (32-bit profiles: the run is linear)
    const uint32_t *in32 = reinterpret_cast<const uint32_t *>(cube);

    // ---- phase 1: heads -> chunk offsets -> gather planes -> inverse transpose -----------------------------
    if constexpr (B == 32) {
        const uint32_t head = in32[t];
        const uint32_t cnt = static_cast<uint32_t>(__builtin_popcount(head));
        const uint32_t incl = wave_inclusive_scan(cnt, lane);
        if (lane == 63) xchg[wave] = incl;
        __syncthreads();
        const uint32_t base = P::head_words + (wave ? xchg[0] : 0u) + incl - cnt;
        // all 32 planes present at a 16-byte aligned position: eight 16-byte reads instead of 32 word reads at a lane
        // stride of 32 words (32-way bank conflicts when whole wavefronts are that dense: incompressible data)
        if (head == 0xffffffffu && ((reinterpret_cast<uintptr_t>(in32 + base) & 15u) == 0)) {
            const char *src = reinterpret_cast<const char *>(in32 + base);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const vec16 v = lds_read16(src + 16 * i);
#pragma unroll
                for (int j = 0; j < 4; ++j) r[4 * i + j] = v.w[j];
            }
            if constexpr (ComplementInPlaneDomain) {
#pragma unroll
                for (int i = 1; i < 32; ++i) r[i] ^= r[0];
            }
        } else {
            // walk the present planes from the LAST one back to the first (as the 64-bit profiles do below): the byte address
            // steps down one word per set head bit -- v_bfe_i32 (0 / -1 per plane) and v_lshl_add_u32 p, kept, 2, p -- and the
            // word under it is read speculatively (inside the run, or the one word behind it, which the staging region holds)
            // and kept iff the bit is set: three VALU instructions per plane with the fused mask + complement below, where the
            // forward walk (shift, and 4, add; mask again for the select) took five.  The 32 masks stay in registers: the
            // decoder's occupancy is bound by its LDS (4 wavefronts per SIMD), not by them.
            uint32_t p = lds_address(in32 + base + cnt);
            int32_t kept[32];
#pragma unroll
            for (int i = 31; i >= 0; --i) {
                kept[i] = opaque_vgpr(static_cast<int32_t>(head << i) >> 31);
                p = static_cast<uint32_t>(opaque_vgpr(static_cast<int32_t>(p + 4 * kept[i])));  // (one v_lshl_add_u32; not a running count)
                r[i] = *reinterpret_cast<const uint32_t *>(lds_pointer(p));
            }
            lds_reads_issued_before_use(r);
            // keep-mask and plane-domain complement in ONE three-input bit operation per plane, (word & kept) ^ sign plane
            // (v_bitop3_b32; done behind the join of the two paths it was an AND here and an XOR there)
            r[0] &= static_cast<uint32_t>(kept[0]);
#pragma unroll
            for (int i = 1; i < 32; ++i) {
                r[i] = ComplementInPlaneDomain ? (r[i] & static_cast<uint32_t>(kept[i])) ^ r[0] : r[i] & static_cast<uint32_t>(kept[i]);
            }
        }
        transpose32(r);
    } else {
        const bool upper = (t & 1) == 0;
        const uint32_t half = upper ? 1u : 0u;
        const uint32_t c = static_cast<uint32_t>(t >> 1);
        const uint32_t head_lo = *R::ptr(in32 + 2 * c), head_hi = *R::ptr(in32 + 2 * c + 1);
        const uint32_t cnt = static_cast<uint32_t>(__builtin_popcount(head_lo) + __builtin_popcount(head_hi));
        const uint32_t incl = wave_inclusive_scan(upper ? cnt : 0u, lane);
        if (lane == 63) xchg[wave] = incl;
        __syncthreads();
        const uint32_t base = P::head_words + (wave ? xchg[0] : 0u) + incl - cnt;
        // walk the chunk's kept planes from the LAST one back to the first: the byte pointer steps down one 64-bit plane per set
        // head bit (pointer += 8 * mask with mask = 0 / -1 from one v_bfe_i32), the word under it is read speculatively (inside
        // the run, or the 8 bytes behind it that the staging region holds) and kept iff the bit is set -- three VALU
        // instructions per plane half instead of the mask / popcount / address / select of an indexed gather
        uint32_t hi[32], lo[32];
        const uint32_t first = lds_address(cube) + 8 * base;  // linear LDS address of the chunk's first plane word
        // All 64 planes present at a 16-byte aligned position (incompressible data): the chunk is 32 whole 16-byte slots, which
        // the swizzle moves as units -- 32 ds_read_b128 (planes 2k and 2k + 1, both halves; the lane keeps its half) instead of
        // 64 word reads at a lane stride of 512 bytes, which the swizzle spreads over only two slot positions: tools/lds_profile.py
        // prices that gather at 8x its conflict-free cycles (random bits 2D f64: 1 792 of 3 103 LDS-pipe cycles per hypercube).
        if ((head_lo & head_hi) == 0xffffffffu && (first & 15u) == 0) {
#pragma unroll
            for (int k = 0; k < 32; ++k) {
                const vec16 v = lds_read16(lds_pointer(R::at(first + 16u * static_cast<uint32_t>(k))));
                uint32_t(&dst)[32] = k < 16 ? hi : lo;
                dst[2 * (k & 15)] = upper ? v.w[1] : v.w[0];
                dst[2 * (k & 15) + 1] = upper ? v.w[3] : v.w[2];
            }
            if constexpr (ComplementInPlaneDomain) {
                // hi[0] is this lane's half of the sign plane; it covers the same 32 values as all its other plane halves
#pragma unroll
                for (int i = 1; i < 32; ++i) hi[i] ^= hi[0];
#pragma unroll
                for (int i = 0; i < 32; ++i) lo[i] ^= hi[0];
            }
        } else {
            uint32_t p = first + 8 * cnt + 4 * half;  // linear LDS address; the word sits at R::at(p)
            int32_t kept_hi[32], kept_lo[32];  // 0 / -1 per plane (the kernel's occupancy is bound by LDS, not by these registers)
#pragma unroll
            for (int i = 31; i >= 0; --i) {
                kept_lo[i] = opaque_vgpr(static_cast<int32_t>(head_lo << i) >> 31);
                p = static_cast<uint32_t>(opaque_vgpr(static_cast<int32_t>(p + 8 * kept_lo[i])));  // (one v_lshl_add_u32; not a running count)
                lo[i] = *reinterpret_cast<const uint32_t *>(lds_pointer(R::at(p)));
            }
#pragma unroll
            for (int i = 31; i >= 0; --i) {
                kept_hi[i] = opaque_vgpr(static_cast<int32_t>(head_hi << i) >> 31);
                p = static_cast<uint32_t>(opaque_vgpr(static_cast<int32_t>(p + 8 * kept_hi[i])));
                hi[i] = *reinterpret_cast<const uint32_t *>(lds_pointer(R::at(p)));
            }
            lds_reads_issued_before_use(lo);
            lds_reads_issued_before_use(hi);
            // keep-mask and plane-domain complement fused: (word & kept) ^ sign plane = one v_bitop3_b32 per plane half
            hi[0] &= static_cast<uint32_t>(kept_hi[0]);
#pragma unroll
            for (int i = 1; i < 32; ++i) {
                hi[i] = ComplementInPlaneDomain ? (hi[i] & static_cast<uint32_t>(kept_hi[i])) ^ hi[0] : hi[i] & static_cast<uint32_t>(kept_hi[i]);
            }
#pragma unroll
            for (int i = 0; i < 32; ++i) {
                lo[i] = ComplementInPlaneDomain ? (lo[i] & static_cast<uint32_t>(kept_lo[i])) ^ hi[0] : lo[i] & static_cast<uint32_t>(kept_lo[i]);
            }
        }
        transpose32(hi);
        transpose32(lo);
#pragma unroll
        for (int j = 0; j < 32; ++j) r[j] = (static_cast<uint64_t>(hi[j]) << 32) | lo[j];
    }
}

// phases 2+3 of decode: residuals -> complement_negative (unless already done) -> prefix sums along every axis ->
// rotr1 -> global
template<typename T, int Dims, bool Aligned, bool AlreadyComplemented = false>
NDZIP_DEV void inverse_transform_hypercube(typename profile<T, Dims>::word (&r)[vals_per_thread],
        typename profile<T, Dims>::word *__restrict__ out, const grid_geom &gg, uint64_t origin, bool active, char *cube,
        uint32_t *xchg, int t) {
    using P = profile<T, Dims>;
    using W = typename P::word;
    using L = lds_layout<W>;
    const int lane = t & 63, wave = wave_uniform(t >> 6);
    if constexpr (!AlreadyComplemented) {
#pragma unroll
        for (int j = 0; j < 32; ++j) r[j] = complement_negative(r[j]);
    }

    // ---- phase 2: prefix sums that stay inside the work-item / wavefront ----------------------------------
    if constexpr (Dims == 1) {
#pragma unroll
        for (int j = 1; j < 32; ++j) r[j] += r[j - 1];
        const W incl = wave_inclusive_scan_w(r[31], lane);
        W *xw = reinterpret_cast<W *>(xchg + 2);  // 2 words of W after the two uint32
        __syncthreads();                           // xchg[0..1] reads above are complete
        if (lane == 63) xw[wave] = incl;
        __syncthreads();
        const W carry = incl - r[31] + (wave ? xw[0] : W{0});
#pragma unroll
        for (int j = 0; j < 32; ++j) r[j] += carry;
    } else if constexpr (Dims == 2) {
#pragma unroll
        for (int j = 1; j < 32; ++j) r[j] += r[j - 1];
        // (the left half row's total from lane t - 1: a DPP row shift -- an odd lane's predecessor is always in its own 16-lane row --
        // where __shfl_up was a ds_bpermute_b32 round trip through the LDS crossbar)
        const W left = group8_shift_up<1>(r[31], ~0u);
        if (t & 1) {
#pragma unroll
            for (int j = 0; j < 32; ++j) r[j] += left;
        }
    } else {
        // x within each row, then y: second row += first row, scan of row pairs over the 8 lanes of a z-plane
#pragma unroll
        for (int j = 1; j < 16; ++j) {
            r[j] += r[j - 1];
            r[16 + j] += r[16 + j - 1];
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) r[16 + j] += r[j];
        const int yp = t & 7;
        // (the masks as opaque register values: hipcc then folds the row shift into the AND -- v_and_b32_dpp -- where a select on
        // the comparison made it v_mov_b32_dpp + v_cndmask: 48 instructions less per work-item)
        const uint32_t keep1 = static_cast<uint32_t>(opaque_vgpr(yp >= 1 ? -1 : 0)), keep2 = static_cast<uint32_t>(opaque_vgpr(yp >= 2 ? -1 : 0));
#pragma unroll
        for (int j = 0; j < 16; ++j) r[16 + j] += group8_shift_up<1>(r[16 + j], keep1);
#pragma unroll
        for (int j = 0; j < 16; ++j) r[16 + j] += group8_shift_up<2>(r[16 + j], keep2);
#pragma unroll
        for (int j = 0; j < 16; ++j) r[16 + j] += group8_shift_up<4>(r[16 + j], 0u);
        // rows of this lane: the first row gets the inclusive total of the previous lane; the second row already
        // holds the inclusive total of this lane
#pragma unroll
        for (int j = 0; j < 16; ++j) r[j] += group8_shift_up<1>(r[16 + j], keep1);
    }

    __syncthreads();  // every work-item has consumed the encoded run: overwrite `cube` with values
    // Where the decoded values wait for the store pass.  1D / 2D and the 64-bit profiles: lds_layout (16 bytes of padding per 32
    // values).  3D f32: UNPADDED 128-byte rows -- row t = the 32 values of work-item t -- with the 16-byte slot s of row r kept at
    // slot s ^ (r & 7) (region-relative: `cube` need not be 128-byte aligned; the same idea as wide::value_layout).  The padded
    // layout cost the store pass below a 2-way conflict on every read: its half-wavefront (y = 4 m .. 4 m + 3, 8 bytes per lane) spans
    // 272 bytes, 16 more than the 64 banks hold (tools/lds_profile.py: 2.00x, 64 of ~1 000 LDS cycles per hypercube).  Here the
    // four rows-of-the-cube of a half-wavefront are two adjacent 128-byte LDS rows, every slot of both touched exactly once (1.00x);
    // the writer's 8 consecutive lanes put slot i at i ^ 0..7 of 8 different rows (1.00x as before).  Cost: one v_xad_u32 per
    // 16-byte write for the slot (8 per work-item); the reader keeps one per-lane address and the plane in the offset field,
    // because a plane is 8 rows and r & 7 does not depend on z.
    constexpr bool swizzled_rows = Dims == 3 && sizeof(W) == 4;
    if constexpr (swizzled_rows) {
        char *row = cube + 128u * static_cast<uint32_t>(t);
        const uint32_t key = (static_cast<uint32_t>(t) & 7u) * 16u;
#pragma unroll
        for (uint32_t i = 0; i < 8; ++i) {
            vec16 v;
#pragma unroll
            for (int j = 0; j < 4; ++j) v.w[j] = static_cast<uint32_t>(r[4 * i + j]);
            lds_write16(row + (key ^ (16u * i)), v);
        }
    } else {
        char *own = cube + L::off(static_cast<uint32_t>(t) * 32u);
        write_run16<W>(own, *reinterpret_cast<W(*)[16]>(&r[0]));
        write_run16<W>(own + 16 * sizeof(W), *reinterpret_cast<W(*)[16]>(&r[16]));
    }
    __syncthreads();

    // ---- phase 3: remaining axis sums in the store layout, rotr1, coalesced global store ---------------------
    if (!active) return;
    if constexpr (Dims == 1) {
        {
            constexpr int VE = 16 / sizeof(W);
            constexpr int NV = hc_size / VE / threads_per_hc;
            vec16 all[NV];  // (every read issued before the first store: one LDS round trip instead of NV)
#pragma unroll
            for (int i = 0; i < NV; ++i) all[i] = lds_read16(cube + L::off(static_cast<uint32_t>(i * threads_per_hc + t) * VE));
            // global address = (wave-uniform: the hypercube's origin + i x 2 KiB, scalar arithmetic) + (one 32-bit per-lane offset)
            char *dst = reinterpret_cast<char *>(scalar_pointer(out + origin));
            const uint32_t lane_bytes = lane_offset_here(static_cast<uint32_t>(t) * 16u);
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                vec16 v = all[i];
                if constexpr (sizeof(W) == 4) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) v.w[j] = rotr1(v.w[j]);
                } else {
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const uint64_t x = rotr1(static_cast<uint64_t>(v.w[2 * j]) | (static_cast<uint64_t>(v.w[2 * j + 1]) << 32));
                        v.w[2 * j] = static_cast<uint32_t>(x);
                        v.w[2 * j + 1] = static_cast<uint32_t>(x >> 32);
                    }
                }
                global_store16<Aligned>(dst + lane_bytes, v);
                dst = scalar_pointer(dst + threads_per_hc * 16);
            }
        }
    } else if constexpr (Dims == 2) {
        // work-item = column x of one half of the rows; the second half first accumulates the first half.
        // A row is 64 values = two padded chunks, so value (y, x) sits at off(x) + y * off(64): ONE per-lane LDS address with the
        // row in the instruction's offset field, ONE running scalar pointer for the hypercube's rows in global memory and one
        // per-lane offset register (the loop used to rebuild all three per row: ~10 VALU instructions and two v_readfirstlane
        // for each of its 32 rows -- the executed-instruction profile showed the 2D f32 decoder at 1 801 VALU per hypercube
        // against 1 358 / 1 237 for 3D / 1D).
        const uint32_t x = static_cast<uint32_t>(lane);
        constexpr uint32_t row_bytes = L::off(64);
        const uint32_t y0 = wave ? 32u : 0u;  // (wave-uniform)
        const char *src = cube + L::off(x);
        W acc = 0;
        if (wave) {
#pragma unroll 8
            for (uint32_t y = 0; y < 32; ++y) acc += lds_read<W>(src, y * row_bytes);
        }
        src += y0 * row_bytes;
        char *dst = reinterpret_cast<char *>(scalar_pointer(out + origin + static_cast<uint64_t>(y0) * gg.stride[0]));
        const uint64_t row_step = gg.stride[0] * sizeof(W);
        const uint32_t lane_bytes = lane_offset_here(x * static_cast<uint32_t>(sizeof(W)));
#pragma unroll 8
        for (uint32_t y = 0; y < 32; ++y) {
            acc += lds_read<W>(src, y * row_bytes);
            *reinterpret_cast<W *>(dst + lane_bytes) = rotr1(acc);
            dst = scalar_pointer(dst + row_step);
        }
    } else {
        // work-item = (y, pair of x) for all 16 z
        const uint32_t y = static_cast<uint32_t>(t) >> 3, xp = static_cast<uint32_t>(t) & 7u;
        W acc0 = 0, acc1 = 0;
        // one LDS base with immediate offsets (a plane is 8 padded chunks);
        // global address = (wave-uniform: hypercube origin + z planes, a running scalar pointer) + (32-bit per-lane byte offset
        // inside a plane: row y, values 2 xp .. 2 xp + 1) -- no 64-bit per-lane pointer advanced by 64-bit VALU additions
        const uint32_t lane_bytes = (y * static_cast<uint32_t>(gg.stride[1]) + 2 * xp) * static_cast<uint32_t>(sizeof(W));
        const uint64_t plane_step = gg.stride[0] * sizeof(W);
        char *dst = reinterpret_cast<char *>(scalar_pointer(out + origin));
        // f32: values (z, y, 2 xp .. 2 xp + 1) = row 8 z + y / 2 of the swizzled layout above, slot 4 (y & 1) + xp / 2, half xp & 1
        const char *src = swizzled_rows ? cube + 128u * (y >> 1) + ((((y & 1u) * 4u + (xp >> 1)) ^ (y >> 1)) * 16u) + (xp & 1u) * 8u
                                        : cube + L::off(y * 16 + 2 * xp);
        constexpr uint32_t plane_bytes = swizzled_rows ? 8u * 128u : L::off(256);
        if constexpr (sizeof(W) == 4) {
#pragma unroll
            for (uint32_t z = 0; z < 16; ++z) {
                const uint2 v = *reinterpret_cast<const uint2 *>(src + z * plane_bytes);
                acc0 += v.x;
                acc1 += v.y;
                global_store8<Aligned>(dst + lane_offset_here(lane_bytes), rotr1(acc0), rotr1(acc1));
                dst = scalar_pointer(dst + plane_step);
            }
        } else {
            // all 16 reads first: written as one loop, hipcc serialises "ds_read_b128, s_waitcnt lgkmcnt(0), add, store" 16 times
            // (the f64 decoder runs at 2 waves/SIMD -- there is little else to cover an LDS round trip with)
            vec16 v[16];
#pragma unroll
            for (uint32_t z = 0; z < 16; ++z) v[z] = lds_read16(src + z * plane_bytes);
#pragma unroll
            for (uint32_t z = 0; z < 16; ++z) {
                acc0 += static_cast<uint64_t>(v[z].w[0]) | (static_cast<uint64_t>(v[z].w[1]) << 32);
                acc1 += static_cast<uint64_t>(v[z].w[2]) | (static_cast<uint64_t>(v[z].w[3]) << 32);
                const uint64_t o0 = rotr1(acc0), o1 = rotr1(acc1);
                vec16 w;
                w.w[0] = static_cast<uint32_t>(o0);
                w.w[1] = static_cast<uint32_t>(o0 >> 32);
                w.w[2] = static_cast<uint32_t>(o1);
                w.w[3] = static_cast<uint32_t>(o1 >> 32);
                global_store16<Aligned>(dst + lane_offset_here(lane_bytes), w);
                dst = scalar_pointer(dst + plane_step);
            }
        }
    }
}

}  // namespace ndzip_hip

namespace ndzip_hip {

template<typename T, int Dims, bool Aligned>
NDZIP_DEV void decode_hypercube(typename profile<T, Dims>::word *__restrict__ out, const grid_geom &gg, uint64_t origin,
        bool active, char *cube, uint32_t run_off, uint32_t *xchg, int t) {
    // `run_off`: byte offset of the encoded run inside `cube` (run_layout; consumed before `cube` is overwritten with values)
    typename profile<T, Dims>::word r[vals_per_thread];
    decode_residuals<T, Dims, true>(cube, run_off, xchg, t, r);
    inverse_transform_hypercube<T, Dims, Aligned, true>(r, out, gg, origin, active, cube, xchg, t);
}

}  // namespace ndzip_hip
