// ndzip_amd/csrc/codec_kernels_wide.hpp -- f64 ENCODE with 256 work-items per hypercube ("wide" mapping): a work-item owns
// 16 consecutive cube-local values, a chunk of 64 values spans a lane quad.
//
// Why: with the 128-work-item mapping of codec_kernels.hpp an f64 work-item carries 64 VGPRs of residuals and 64 VGPRs of
// plane words -- too much to keep an encoded tile in registers across an iteration, so an f64 compress kernel of that shape
// has to write a tile out in the iteration that encoded it, with its look-back on the critical path.  Here a work-item
// carries, per 16 values, 32 VGPRs of values, the same of prefetch and -- after the transpose -- the same of plane words:
// exactly the register picture of the 128-lane f32 kernel, so its register-buffered deferred-write-out pipeline carries over
// (2D f64 8192^2: 0.253 -> 0.210 ms; 3D f64 512^3: 0.479 -> 0.391 ms).  (The same mapping for f32 -- one 16 KiB hypercube
// per tile -- doubles the tiles, i.e. tickets, descriptors and look-backs, and measured 0.284 vs 0.208 ms on 512^3; it is
// not kept.)
//
// Bit transpose of a chunk over its lane quad q = 0..3 (values 16q .. 16q+15): the two lanes of a pair swap halves -- the
// even lane ends up with the HIGH dwords of the pair's 32 values, the odd lane with the LOW dwords (16 DPP moves) -- then one
// 32x32 transpose per lane.  Lane 0 holds, for planes 0..31 (bits 63..32), the dword covering values 0..31 = the HIGH dword
// of the 64-bit plane word; lane 2 the LOW dword of the same planes (values 32..63); lanes 1 / 3 likewise for planes 32..63.
// Each lane compacts its own 32 plane words (conditional LDS writes).
//
// Reference behaviour restated here (not its structure): load_hypercube + rotate_left_1 (src/ndzip/cuda_codec.inl:30-56,
// common.hh:436-440), block_transform (cuda_codec.inl:68-126; the per-axis passes are fused into one stencil),
// complement_negative (common.hh:442-449), write_transposed_chunks = chunk head, BxB bit transpose, zero-word compaction
// (cuda_codec.inl:185-275; stream layout common.hh:328-366).  Parity: every float64 compress test (tests/test_hip_*.py on the
// GPU, the functional-model tests on the CPU) runs through this mapping; stage tests call these functions one at a time.
#pragma once

#include "codec_kernels.hpp"

namespace ndzip_hip {
namespace wide {

constexpr int threads = 256;  // work-items per hypercube
constexpr int vals = 16;      // cube-local values [16 t, 16 t + 16) per work-item

template<typename W>
struct layout {
    static constexpr uint32_t chunk_bytes = vals * sizeof(W) + 16;  // 144 (f64: 9 slots of 16 bytes) / 80 (f32: 5 slots)
    static constexpr uint32_t cube_bytes = threads * chunk_bytes;    // 36864 / 20480
    static constexpr uint32_t zero_bytes = vals * sizeof(W) + 256;
    // The 16-byte slot (mod 16) that no in-cube lane of a 16-lane group touches in the neighbour reads.  Lane t sits in slot
    // 9t (f64) / 5t (f32) mod 16; the border lanes of rows y-1 (lane t-1, 3D), (z-1, y-1) (t-17) and of the 2D row above
    // (t-4) would have used slot 7 (f64) / 11 (f32).
    static constexpr uint32_t zero_offset = (sizeof(W) == 8 ? 7 : 11) * 16;
    NDZIP_DEV static constexpr uint32_t off(uint32_t k) { return k * static_cast<uint32_t>(sizeof(W)) + (k >> 4) * 16u; }
};

template<typename W>
struct input_regs {
    static constexpr int VE = 16 / sizeof(W);
    static constexpr int NV = hc_size / VE / threads;  // 8 (f64) / 4 (f32) vectors
    vec16 v[NV];
};

// coalesced global loads: vector i of work-item t covers values (i*256 + t)*VE; 256*VE values are whole rows / planes
template<typename W, int Dims, bool Aligned, int Part = -1, int Split = 0>
NDZIP_DEV void load_regs(const W *__restrict__ in, const grid_geom &gg, uint64_t origin, int t, input_regs<W> &regs) {
    using R = input_regs<W>;
    constexpr int first = Part == 1 ? Split : 0;
    constexpr int last = Part == 0 ? Split : R::NV;
    // (wave-uniform base + 32-bit per-lane byte offset: see load_hypercube_regs)
    uint64_t plane;
    uint32_t lane_bytes;
    split_local_offset<Dims, W>(gg, static_cast<uint32_t>(t) * R::VE, plane, lane_bytes);
    const char *base = reinterpret_cast<const char *>(in + origin + plane);
    const uint64_t step = local_offset<Dims>(gg, threads * R::VE) * sizeof(W);
    const uint32_t off = lane_offset_here(lane_bytes);
    const char *p = scalar_pointer(base + first * step);  // (a running scalar pointer: see load_hypercube_regs)
#pragma unroll
    for (int i = first; i < last; ++i) {
        regs.v[i] = global_load16<Aligned>(p + off);
        if (i + 1 < last) p = scalar_pointer(p + step);
    }
}

template<typename W>
NDZIP_DEV void stage_regs(const input_regs<W> &regs, char *cube, int t) {
    using R = input_regs<W>;
    using L = layout<W>;
    char *base = cube + L::off(static_cast<uint32_t>(t) * R::VE);
    constexpr uint32_t step = L::off(threads * R::VE);
#pragma unroll
    for (int i = 0; i < R::NV; ++i) {
        vec16 r;
        if constexpr (sizeof(W) == 4) {
#pragma unroll
            for (int j = 0; j < 4; ++j) r.w[j] = rotl1(regs.v[i].w[j]);
        } else {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const uint64_t x = rotl1(static_cast<uint64_t>(regs.v[i].w[2 * j]) | (static_cast<uint64_t>(regs.v[i].w[2 * j + 1]) << 32));
                r.w[2 * j] = static_cast<uint32_t>(x);
                r.w[2 * j + 1] = static_cast<uint32_t>(x >> 32);
            }
        }
        lds_write16(base + i * step, r);
    }
}

// fused Lorenzo stencil out of LDS + complement_negative -> residuals r[16] of work-item t
template<typename W, int Dims>
NDZIP_DEV void stencil(const char *cube, const char *zero, int t, W (&r)[vals]) {
    using L = layout<W>;
    const uint32_t k0 = static_cast<uint32_t>(t) * vals;
    const char *own = cube + L::off(k0);
    constexpr int Q = 32 / sizeof(W);  // values folded at a time (two 16-byte reads per row)
    if constexpr (Dims == 1) {
        W prev = lds_read<W>(t > 0 ? cube + L::off(k0 - 1) : zero, 0);
#pragma unroll
        for (int q = 0; q < vals / Q; ++q) {
            W o[Q];
            read_run<W, Q>(own + q * Q * sizeof(W), o);
#pragma unroll
            for (int j = Q - 1; j >= 1; --j) r[q * Q + j] = o[j] - o[j - 1];
            r[q * Q] = o[0] - prev;
            prev = o[Q - 1];
            __builtin_amdgcn_sched_barrier(0);
        }
    } else if constexpr (Dims == 2) {
        // 64 x 64: work-item = quarter row; y = t / 4, quarter = t % 4
        const int y = t >> 2, qx = t & 3;
        const char *up = y > 0 ? cube + L::off(k0 - 64) : zero;
        const W ol = lds_read<W>(qx ? cube + L::off(k0 - 1) : zero, 0);
        const W ul = lds_read<W>((qx && y > 0) ? cube + L::off(k0 - 65) : zero, 0);
        W left = ol - ul;
#pragma unroll
        for (int q = 0; q < vals / Q; ++q) {
            W o[Q], u[Q];
            read_run<W, Q>(own + q * Q * sizeof(W), o);
            read_run<W, Q>(up + q * Q * sizeof(W), u);
#pragma unroll
            for (int j = 0; j < Q; ++j) o[j] -= u[j];
#pragma unroll
            for (int j = Q - 1; j >= 1; --j) r[q * Q + j] = o[j] - o[j - 1];
            r[q * Q] = o[0] - left;
            left = o[Q - 1];
            __builtin_amdgcn_sched_barrier(0);
        }
    } else {
        // 16^3: work-item = row (z, y) = (t / 16, t % 16)
        const int z = t >> 4, y = t & 15;
        const char *row_p = y > 0 ? cube + L::off(k0 - 16) : zero;                    // (z, y-1)
        const char *row_a1 = z > 0 ? cube + L::off(k0 - 256) : zero;                 // (z-1, y)
        const char *row_p1 = (z > 0 && y > 0) ? cube + L::off(k0 - 256 - 16) : zero;  // (z-1, y-1)
        W carry = 0;
#pragma unroll
        for (int q = 0; q < vals / Q; ++q) {
            const uint32_t o = q * Q * sizeof(W);
            W a[Q];
            read_run<W, Q>(own + o, a);
            {
                W p[Q];
                read_run<W, Q>(row_p + o, p);
#pragma unroll
                for (int j = 0; j < Q; ++j) a[j] -= p[j];
            }
            __builtin_amdgcn_sched_barrier(0);
            {
                W a1[Q], p1[Q];
                read_run<W, Q>(row_a1 + o, a1);
                read_run<W, Q>(row_p1 + o, p1);
#pragma unroll
                for (int j = 0; j < Q; ++j) a[j] -= a1[j] - p1[j];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = Q - 1; j >= 1; --j) r[q * Q + j] = a[j] - a[j - 1];
            r[q * Q] = a[0] - carry;
            carry = a[Q - 1];
        }
    }
#pragma unroll
    for (int j = 0; j < vals; ++j) r[j] = complement_negative(r[j]);
}

// value held by the other lane of the pair (t ^ 1): DPP quad_perm [1,0,3,2]
NDZIP_DEV uint32_t pair_swap(uint32_t v) {
    return static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0xb1, 0xf, 0xf, true));
}
// OR over the 4 lanes of a quad
NDZIP_DEV uint32_t quad_or(uint32_t v) {
    v |= static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0xb1, 0xf, 0xf, true));  // [1,0,3,2]
    v |= static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x4e, 0xf, 0xf, true));  // [2,3,0,1]
    return v;
}

// What one work-item contributes to the encoded run of its chunk.
template<typename W>
struct coding;

template<>
struct coding<uint64_t> {
    static constexpr int lanes_per_chunk = 4;
    static constexpr int planes_per_lane = 32;
    static constexpr uint32_t head_words = hc_size / 64;  // in 64-bit words
    static constexpr uint32_t w32 = 2;                     // uint32 per stream word
    // state a lane keeps for the deferred write-out
    struct held {
        uint32_t head_bits;  // head bits of this lane's planes, MSB = its first plane
        uint32_t head_word;  // the uint32 of the chunk head this lane stores (lanes 0 / 1: high / low dword)
        uint32_t slot;       // uint32 index of this lane's first plane word inside the run
    };
    // chunk head (same on the 4 lanes), count of its planes
    NDZIP_DEV static uint32_t head_and_count(const uint64_t (&r)[vals], uint32_t &head_hi, uint32_t &head_lo) {
        uint64_t own = 0;
#pragma unroll
        for (int j = 0; j < vals; ++j) own |= r[j];
        head_hi = quad_or(static_cast<uint32_t>(own >> 32));
        head_lo = quad_or(static_cast<uint32_t>(own));
        return static_cast<uint32_t>(__builtin_popcount(head_hi) + __builtin_popcount(head_lo));
    }
    NDZIP_DEV static void transpose(const uint64_t (&r)[vals], int t, uint32_t (&planes)[planes_per_lane]) {
        // the even lane sends its low dwords, the odd lane its high ones: planes[j] (values 0..15 of the pair, the even lane's) =
        // odd ? the even lane's low dword : own high dword; planes[16 + j] (the odd lane's values) = odd ? own low dword : the odd
        // lane's high dword -- swap and choice in one v_cndmask_b32_dpp per dword (gfx950_lds.hpp: pair_exchange_select4)
        const uint32_t odd_flag = static_cast<uint32_t>(t & 1);
#pragma unroll
        for (int j = 0; j < vals; j += 4) {
            const uint32_t hi[4] = {static_cast<uint32_t>(r[j] >> 32), static_cast<uint32_t>(r[j + 1] >> 32), static_cast<uint32_t>(r[j + 2] >> 32),
                    static_cast<uint32_t>(r[j + 3] >> 32)};
            const uint32_t lo[4] = {static_cast<uint32_t>(r[j]), static_cast<uint32_t>(r[j + 1]), static_cast<uint32_t>(r[j + 2]), static_cast<uint32_t>(r[j + 3])};
            uint32_t own_half[4], other_half[4];
            pair_exchange_select4(odd_flag, hi, lo, own_half, other_half);  // (lo_out = odd ? lo : swap(hi); hi_out = odd ? swap(lo) : hi)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                planes[j + k] = other_half[k];
                planes[vals + j + k] = own_half[k];
            }
        }
        transpose32(planes);
    }
    // chunk_pos: 64-bit word index of the chunk's first plane word inside the run
    NDZIP_DEV static held hold(int t, uint32_t head_hi, uint32_t head_lo, uint32_t chunk_pos) {
        const int q = t & 3;
        held h;
        h.head_bits = (q & 1) ? head_lo : head_hi;
        h.head_word = (q & 1) ? head_lo : head_hi;
        const uint32_t pos = chunk_pos + ((q & 1) ? static_cast<uint32_t>(__builtin_popcount(head_hi)) : 0u);
        h.slot = 2 * pos + ((q & 2) ? 0u : 1u);  // lanes 0, 1: high dword of the plane word; lanes 2, 3: low dword
        return h;
    }
    // `run32`: start of the run's LDS region (the run begins at its first byte); words go where run_layout<uint64_t> puts them
    NDZIP_DEV static void write(const held &h, const uint32_t (&planes)[planes_per_lane], uint32_t *run32, int t) {
        using R = run_layout<uint64_t>;
        const int q = t & 3;
        const uint32_t c = static_cast<uint32_t>(t) >> 2;
        if (q < 2) *R::ptr(run32 + 2 * c + (q == 0 ? 1u : 0u)) = h.head_word;
        // Dense wavefront (every chunk of it keeps all 64 planes, at an even word position: incompressible data): the compaction
        // below would store 32 dwords per lane at a lane stride of 256 bytes -- with the swizzle still an 8-way bank conflict on
        // each of its 32 instructions.  Instead the two lanes that hold the halves of the same planes (t, t ^ 2: high / low
        // dwords) trade halves -- the lower lane takes planes 0..15 of its 32, the upper one 16..31 -- and every lane writes its
        // 16 whole 64-bit words as eight 16-byte stores (two DPP moves + two selects per word pair; compiled code, no assembly).
        // Wave-uniform: one s_cmp on a ballot.  (What the f32 encoder's write_planes32 does per chunk.)
        if (__ballot(h.head_bits != 0xffffffffu || ((h.slot >> 1) & 1u) != 0) == 0) {
            const bool upper = (q & 2) != 0;
            const uint32_t a = lds_address(run32) + 8u * ((h.slot >> 1) + (upper ? 16u : 0u));  // (linear; 16-byte aligned)
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                vec16 v;
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int i = 2 * k + e;
                    const uint32_t others_first = static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(planes[i]), 0x4e, 0xf, 0xf, true));
                    const uint32_t others_second = static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(planes[16 + i]), 0x4e, 0xf, 0xf, true));
                    v.w[2 * e] = upper ? planes[16 + i] : others_first;       // low dword: lanes 2, 3 hold it
                    v.w[2 * e + 1] = upper ? others_second : planes[i];       // high dword: lanes 0, 1 hold it
                }
                lds_write16(lds_pointer(R::at(a + 16u * static_cast<uint32_t>(k))), v);
            }
            return;
        }
        // a running linear LDS address, one 64-bit stream word per kept plane; branch-free, EXEC-masked (gfx950_lds.hpp), with
        // run_layout's swizzle folded into the store address
        static_assert(R::at(0x3f8u) == (0x3f8u ^ 0x70u), "lds_append_flagged64 spells run_layout<uint64_t>::at out");
        lds_append_flagged64(lds_address(run32 + h.slot), h.head_bits, planes);
    }
};


// =========================================================================================================================
// f64 DECODE with 256 work-items per hypercube (the mirror image of the encoder above; decompress_kernel_wide).
//
// Why: the 128-work-item decoder of codec_kernels.hpp carries, per f64 work-item, 64 plane halves + 64 keep-masks through
// the gather and 32 x 64-bit values through the inverse transform: 146 VGPRs, and with 35 KB of LDS per hypercube and two
// wavefronts per workgroup a CU holds 4 workgroups = 2 wavefronts per SIMD -- nothing to cover an LDS or memory round trip
// with.  Here a work-item gathers 32 plane dwords (one 32x32 transpose instead of two) and carries 16 values: half the
// registers per lane, the same LDS per hypercube, twice the wavefronts per CU (4 per SIMD).
//
// Lane roles inside a chunk's quad q = t % 4 (exactly what coding<uint64_t>::hold gave the encoder's lanes): q & 1 picks the
// planes -- 0..31 (head_hi) or 32..63 (head_lo) --, q & 2 the dword of the 64-bit plane word -- lanes 0, 1 the HIGH dword
// (values 0..31 of the chunk), lanes 2, 3 the LOW dword (values 32..63).
//
// Reference behaviour restated (not its structure): read_transposed_chunks / zero-word expansion + inverse bit transpose
// (src/ndzip/cuda_codec.inl:278-365, cpu_codec.inl:561-578), complement_negative (common.hh:442-449),
// inverse_block_transform (cuda_codec.inl:129-183, common.hh:493-535), rotate_right_1 + store_hypercube
// (common.hh:436-440, cuda_codec.inl:58-65).
// =========================================================================================================================

// LDS of decompress_kernel_wide: [region_bytes: the encoded run, later the decoded values (value_layout below)]
// [4 x uint32 wave totals][4 x uint64 1D wave carries][2D: 3 x 64 column totals of the row quarters].  No zero block.
template<int Dims>
struct decode_layout {
    // run + worst misalignment + the plane word read behind it, in whole 128-byte swizzle blocks (>= the 32 KiB of values)
    static constexpr uint32_t region_bytes = ((hc_size + hc_size / 64 + 1) * 8 + 16 + 127) / 128 * 128 + 128;
    static_assert(region_bytes >= hc_size * 8, "the decoded values fit the region");
    static constexpr uint32_t totals_offset = region_bytes;            // uint32[4]
    static constexpr uint32_t carries_offset = totals_offset + 16;     // uint64[4]
    static constexpr uint32_t columns_offset = carries_offset + 32;    // uint64[3 * 64] (2D only)
    static constexpr uint32_t smem_bytes = columns_offset + (Dims == 2 ? 3 * 64 * 8 : 0) + 16;
};

// Where the DECODED values wait between the in-register sums and the store pass: unpadded, 128-byte row r = values
// [16 r, 16 r + 16) with its 16-byte slot s at slot s ^ (r & 7).  Checked against the guide's LDS table (lane groups and bank
// windows per access width; tools/lds_profile.py prices it at 1.00x):
//   * the writer (work-item t = row t, eight ds_write_b128): served 8 consecutive lanes at a time over a 128-byte window -- the 8
//     rows of a group put slot i at i ^ 0..7, all different (padded like lds_layout, 16 bytes per 32 values, lanes 2m / 2m + 1
//     met in one slot: every write a 2-way conflict; the pad per row of wide::layout avoids that but breaks the reads below);
//   * the 3D reader (lane = (y, x), ds_read_b64 of row 16 z + y): 32 lanes = rows y, y + 1 = the two halves of a 256-byte
//     window, 16 lanes each over the row's 8 slots x 2;  the 2D reader (lane = column x): 32 lanes = two quarter rows, likewise;
//   * the 1D reader (lane t reads slot t % 8 of row t / 8 + 32 i, ds_read_b128): the table's 16-lane groups {0-3, 12-15, 20-27}
//     ... see rows r, r + 1, r + 2, r + 3 at slots {0-3}, {4-7} ^ 1, {4-7} ^ 2, {0-3} ^ 3 in alternating halves: 16 different.
// On LDS byte ADDRESSES as integers (gfx950_lds.hpp), like run_layout: one v_xor per access whose slot is not a constant.
struct value_layout {
    // LDS address of slot 0 of row `r` with the row's swizzle folded in: the 16-byte slot s lives at  row_base(r) ^ (16 s)
    NDZIP_DEV static uint32_t row_base(uint32_t region, uint32_t r) { return (region + 128u * r) | (16u * (r & 7u)); }
};

// phase 1: encoded run (run_layout<uint64_t>, `run_off` bytes into `region`) -> the 16 residuals of work-item t, with
// complement_negative already undone (in the plane domain: every plane below the sign plane XOR the sign plane).
// Contains one __syncthreads().
NDZIP_DEV void decode_residuals(const char *region, uint32_t run_off, uint32_t *totals, int t, uint64_t (&r)[vals]) {
    using R = run_layout<uint64_t>;
    constexpr uint32_t head_words = hc_size / 64;
    const int lane = t & 63, wave = wave_uniform(t >> 6);
    const int q = t & 3;
    const bool odd = (q & 1) != 0;
    const uint32_t c = static_cast<uint32_t>(t) >> 2;
    const char *run = region + run_off;
    const uint32_t *in32 = reinterpret_cast<const uint32_t *>(run);
    // (the four lanes of a chunk read the same two head dwords: an LDS broadcast)
    const uint32_t head_lo = *R::ptr(in32 + 2 * c), head_hi = *R::ptr(in32 + 2 * c + 1);
    const uint32_t cnt_hi = static_cast<uint32_t>(__builtin_popcount(head_hi));
    const uint32_t cnt = cnt_hi + static_cast<uint32_t>(__builtin_popcount(head_lo));
    // (the lanes of a chunk end up with the same inclusive value: only the first one feeds the scan)
    const uint32_t incl = wave_inclusive_scan(q == 0 ? cnt : 0u, lane);
    if (lane == 63) totals[wave] = incl;
    __syncthreads();
    uint32_t base = head_words + incl - cnt;
#pragma unroll
    for (int w = 0; w < threads / 64 - 1; ++w) {
        if (w < wave) base += totals[w];
    }
    const uint32_t first = lds_address(run) + 8 * base;   // linear LDS address of the chunk's first plane word
    const uint32_t mine = odd ? head_lo : head_hi;        // head bits of this lane's 32 planes, MSB = its first plane
    const uint32_t half = (q & 2) ? 0u : 4u;              // byte of this lane's dword inside a plane word
    uint32_t w[32];
    uint32_t sign;  // this lane's dword of the chunk's sign plane (plane 0: held by the even lane of the pair)
    // Every chunk of the wavefront keeps all 64 planes at a 16-byte aligned position (incompressible data; all chunks are then
    // 512 bytes long, so they are aligned together): a lane's 32 plane dwords are 16 whole slots of the swizzled run -- 16
    // ds_read_b128 instead of 32 word reads at a lane stride of 512 bytes within a quad's 256, which the swizzle spreads over
    // two slot positions only.  Wave-uniform on purpose: everything else pays one ballot and a scalar branch for it.
    if (__ballot(!((head_lo & head_hi) == 0xffffffffu && (first & 15u) == 0)) == 0) {
        const uint32_t a = first + (odd ? 256u : 0u);
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const vec16 v = lds_read16(lds_pointer(R::at(a + 16u * static_cast<uint32_t>(k))));
            w[2 * k] = half ? v.w[1] : v.w[0];
            w[2 * k + 1] = half ? v.w[3] : v.w[2];
        }
        const uint32_t pair0 = pair_swap(w[0]);  // (every lane executes the swap: a DPP move reads lanes that must be active)
        sign = odd ? pair0 : w[0];
        w[0] ^= odd ? sign : 0u;
#pragma unroll
        for (int i = 1; i < 32; ++i) w[i] ^= sign;
    } else {
        // walk this lane's kept planes from the LAST one back to the first (see decode_residuals in codec_kernels.hpp): the byte
        // address steps down one 64-bit plane word per set head bit, the dword under it is read speculatively (inside the run, or
        // the 8 bytes behind it, which the staging region holds) and kept iff the bit is set
        uint32_t p = first + 8 * (odd ? cnt : cnt_hi) + half;  // linear LDS address; the dword sits at R::at(p)
        int32_t kept[32];
#pragma unroll
        for (int i = 31; i >= 0; --i) {
            kept[i] = opaque_vgpr(static_cast<int32_t>(mine << i) >> 31);
            p = static_cast<uint32_t>(opaque_vgpr(static_cast<int32_t>(p + 8 * kept[i])));  // (one v_lshl_add_u32; not a running count)
            w[i] = *reinterpret_cast<const uint32_t *>(lds_pointer(R::at(p)));
        }
        lds_reads_issued_before_use(w);
        // keep-mask and plane-domain complement fused: (word & kept) ^ sign plane = one v_bitop3_b32 per plane dword.  The odd
        // lane's planes 32..63 are all below the sign plane, which its pair's even lane holds as plane 0.
        const uint32_t own0 = w[0] & static_cast<uint32_t>(kept[0]);
        const uint32_t pair0 = pair_swap(own0);  // (every lane executes the swap: a DPP move reads lanes that must be active)
        sign = odd ? pair0 : own0;
        w[0] = own0 ^ (odd ? sign : 0u);
#pragma unroll
        for (int i = 1; i < 32; ++i) w[i] = (w[i] & static_cast<uint32_t>(kept[i])) ^ sign;
    }
    // inverse of coding<uint64_t>::transpose: one 32x32 transpose per lane -- the even lane then holds the HIGH dwords of the
    // pair's 32 values, the odd lane the LOW dwords -- and the pair swaps halves back (16 DPP moves)
    transpose32(w);
    // even lane: value j = (own w[j], the odd lane's w[j]); odd lane: value 16 + j = (the even lane's w[16 + j], own w[16 + j]):
    // swap and choice in one v_cndmask_b32_dpp per dword (gfx950_lds.hpp: pair_exchange_select4), two instructions per value where
    // "select what to send, DPP move, two selects" took four
    const uint32_t odd_flag = static_cast<uint32_t>(q & 1);
#pragma unroll
    for (int j = 0; j < vals; j += 4) {
        const uint32_t a[4] = {w[j], w[j + 1], w[j + 2], w[j + 3]};
        const uint32_t b[4] = {w[vals + j], w[vals + j + 1], w[vals + j + 2], w[vals + j + 3]};
        uint32_t lo[4], hi[4];
        pair_exchange_select4(odd_flag, a, b, lo, hi);
#pragma unroll
        for (int k = 0; k < 4; ++k) r[j + k] = (static_cast<uint64_t>(hi[k]) << 32) | lo[k];
    }
}

// row_shr:D inside the 16-lane DPP row with zero fill at the row start (a 64-bit value as two dwords)
// (bound_ctrl with every row and bank enabled: the destination's old value is dead, so no register is zeroed for it first)
template<int D>
NDZIP_DEV uint64_t row_shift_up(uint64_t v) {
    const uint32_t lo = static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(static_cast<uint32_t>(v)), 0x110 + D, 0xf, 0xf, true));
    const uint32_t hi = static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(static_cast<uint32_t>(v >> 32)), 0x110 + D, 0xf, 0xf, true));
    return (static_cast<uint64_t>(hi) << 32) | lo;
}

NDZIP_DEV vec16 rotr1_pair(const vec16 &v) {
    vec16 o;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const uint64_t x = rotr1(static_cast<uint64_t>(v.w[2 * j]) | (static_cast<uint64_t>(v.w[2 * j + 1]) << 32));
        o.w[2 * j] = static_cast<uint32_t>(x);
        o.w[2 * j + 1] = static_cast<uint32_t>(x >> 32);
    }
    return o;
}

// phases 2 + 3: residuals (already complemented) -> prefix sums along every axis -> rotr1 -> global.
// `cube`: the region the run was staged into (consumed by now: a barrier inside orders that; 128-byte aligned), `smem`: the
// workgroup's LDS (decode_layout).  Contains barriers: every work-item of the workgroup calls it; `active` guards the stores only.
template<int Dims, bool Aligned>
NDZIP_DEV void inverse_transform(uint64_t (&r)[vals], uint64_t *__restrict__ out, const grid_geom &gg, uint64_t origin, bool active,
        char *cube, char *smem, int t) {
    using W = uint64_t;
    using V = value_layout;
    using D = decode_layout<Dims>;
    const int lane = t & 63, wave = wave_uniform(t >> 6);
    const uint32_t region = lds_address(cube);

    // ---- phase 2: prefix sums that stay inside the work-item / wavefront ----------------------------------
#pragma unroll
    for (int j = 1; j < vals; ++j) r[j] += r[j - 1];  // x (1D: the work-item's 16 consecutive values)
    if constexpr (Dims == 1) {
        W *carries = reinterpret_cast<W *>(smem + D::carries_offset);
        const W incl = wave_inclusive_scan_w(r[vals - 1], lane);
        if (lane == 63) carries[wave] = incl;
        __syncthreads();
        W carry = incl - r[vals - 1];
#pragma unroll
        for (int w = 0; w < threads / 64 - 1; ++w) {
            if (w < wave) carry += carries[w];
        }
#pragma unroll
        for (int j = 0; j < vals; ++j) r[j] += carry;
    } else if constexpr (Dims == 2) {
        // work-item = quarter row (y = t / 4, quarter = t % 4): exclusive scan of the quarters' totals over the quad
        const int qx = t & 3;
        const uint32_t keep1 = static_cast<uint32_t>(opaque_vgpr(qx >= 1 ? -1 : 0)), keep2 = static_cast<uint32_t>(opaque_vgpr(qx >= 2 ? -1 : 0));
        W incl = r[vals - 1];
        incl += group8_shift_up<1>(incl, keep1);
        incl += group8_shift_up<2>(incl, keep2);
        const W left = incl - r[vals - 1];
#pragma unroll
        for (int j = 0; j < vals; ++j) r[j] += left;
    } else {
        // work-item = row (z, y) = (t / 16, t % 16): the 16 rows of a z-plane are the 16 lanes of a DPP row -- y is a plain row
        // scan, no masks (lanes shifted in from outside the row read as 0).  Two instructions per value and step
        // (v_add_co_u32_dpp + v_addc_co_u32_dpp, gfx950_lds.hpp: row_scan_step64) where the compiled form takes three.
        uint32_t lo[2][8], hi[2][8];
#pragma unroll
        for (int j = 0; j < vals; ++j) {
            lo[j >> 3][j & 7] = static_cast<uint32_t>(r[j]);
            hi[j >> 3][j & 7] = static_cast<uint32_t>(r[j] >> 32);
        }
        row_scan_step64<1>(lo[0], hi[0]);
        row_scan_step64<1>(lo[1], hi[1]);
        row_scan_step64<2>(lo[0], hi[0]);
        row_scan_step64<2>(lo[1], hi[1]);
        row_scan_step64<4>(lo[0], hi[0]);
        row_scan_step64<4>(lo[1], hi[1]);
        row_scan_step64<8>(lo[0], hi[0]);
        row_scan_step64<8>(lo[1], hi[1]);
#pragma unroll
        for (int j = 0; j < vals; ++j) r[j] = (static_cast<uint64_t>(hi[j >> 3][j & 7]) << 32) | lo[j >> 3][j & 7];
    }

    __syncthreads();  // every work-item has consumed the encoded run: overwrite `cube` with values
    {
        const uint32_t row = V::row_base(region, static_cast<uint32_t>(t));  // work-item t holds row t
#pragma unroll
        for (int i = 0; i < vals / 2; ++i) {
            vec16 v;
            v.w[0] = static_cast<uint32_t>(r[2 * i]);
            v.w[1] = static_cast<uint32_t>(r[2 * i] >> 32);
            v.w[2] = static_cast<uint32_t>(r[2 * i + 1]);
            v.w[3] = static_cast<uint32_t>(r[2 * i + 1] >> 32);
            lds_write16(lds_pointer(row ^ (16u * static_cast<uint32_t>(i))), v);
        }
    }
    __syncthreads();

    // ---- phase 3: remaining axis sums in the store layout, rotr1, coalesced global store ---------------------
    if constexpr (Dims == 1) {
        constexpr int NV = hc_size / 2 / threads;  // 8 vectors of two values per work-item
        // vector i of work-item t = values (256 i + t) * 2 = slot t % 8 of row 32 i + t / 8: one address, immediate offsets
        const uint32_t src = V::row_base(region, static_cast<uint32_t>(t) >> 3) ^ (16u * (static_cast<uint32_t>(t) & 7u));
        vec16 all[NV];  // (every read issued before the first store: one LDS round trip instead of NV)
#pragma unroll
        for (int i = 0; i < NV; ++i) all[i] = lds_read16(lds_pointer(src) + i * (threads / 8) * 128);
        if (active) {
            // global address = (wave-uniform: the hypercube's origin + i x 4 KiB, scalar arithmetic) + (one 32-bit per-lane offset)
            char *dst = reinterpret_cast<char *>(scalar_pointer(out + origin));
            const uint32_t lane_bytes = lane_offset_here(static_cast<uint32_t>(t) * 16u);
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                global_store16<Aligned>(dst + lane_bytes, rotr1_pair(all[i]));
                dst = scalar_pointer(dst + threads * 16);
            }
        }
    } else if constexpr (Dims == 2) {
        // work-item = column x of one QUARTER of the rows (wavefront k: rows 16 k .. 16 k + 15): column sums of the quarter in
        // registers, the quarters' totals exchanged through LDS -- every wavefront reads 16 rows and stores 16 (the 128-work-item
        // decoder lets its second wavefront re-read the first one's 32 rows instead: 64 dependent reads against 32)
        W *columns = reinterpret_cast<W *>(smem + D::columns_offset);
        const uint32_t x = static_cast<uint32_t>(lane);
        const uint32_t y0 = static_cast<uint32_t>(wave) * 16u;
        // value (y, x) = row 4 y + x / 16, slot (x / 2) % 8, half x % 2; the row's swizzle (4 y + x / 16) % 8 alternates with y % 2
        const uint32_t even = (V::row_base(region, 4 * y0 + (x >> 4)) ^ (16u * ((x >> 1) & 7u))) + 8u * (x & 1u);
        const uint32_t odd = even ^ 64u;
        W v[16];
#pragma unroll
        for (uint32_t j = 0; j < 16; ++j) v[j] = *reinterpret_cast<const W *>(lds_pointer((j & 1u) ? odd : even) + j * 512);
#pragma unroll
        for (int j = 1; j < 16; ++j) v[j] += v[j - 1];
        if (wave < threads / 64 - 1) columns[wave * 64 + lane] = v[15];
        __syncthreads();
        W above = 0;
#pragma unroll
        for (int k = 0; k < threads / 64 - 1; ++k) {
            if (k < wave) above += columns[k * 64 + lane];
        }
        if (active) {
            // (uniform: the hypercube's row y0, a running scalar pointer; per lane: the column)
            char *dst = reinterpret_cast<char *>(scalar_pointer(out + origin + static_cast<uint64_t>(y0) * gg.stride[0]));
            const uint64_t row_step = gg.stride[0] * sizeof(W);
            const uint32_t lane_bytes = lane_offset_here(x * static_cast<uint32_t>(sizeof(W)));  // (once: the stores share a basic block)
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                *reinterpret_cast<W *>(dst + lane_bytes) = rotr1(v[j] + above);
                dst = scalar_pointer(dst + row_step);
            }
        }
    } else {
        // work-item = (y, x) for all 16 z: every wavefront reads 16 x 8 bytes per lane and stores them (four rows of 128 bytes
        // = four whole cache lines per store instruction when the rows are aligned); value (z, y, x) = row 16 z + y, slot x / 2
        const uint32_t y = static_cast<uint32_t>(t) >> 4, x = static_cast<uint32_t>(t) & 15u;
        const uint32_t src = (V::row_base(region, y) ^ (16u * (x >> 1))) + 8u * (x & 1u);
        W v[16];  // (all reads first: see inverse_transform_hypercube)
#pragma unroll
        for (uint32_t z = 0; z < 16; ++z) v[z] = *reinterpret_cast<const W *>(lds_pointer(src) + z * 2048);
        if (active) {
            // global address = (wave-uniform: hypercube origin + z planes, a running scalar pointer) + (32-bit per-lane byte offset
            // inside a plane: row y, value x)
            const uint32_t lane_bytes = lane_offset_here((y * static_cast<uint32_t>(gg.stride[1]) + x) * static_cast<uint32_t>(sizeof(W)));
            const uint64_t plane_step = gg.stride[0] * sizeof(W);
            char *dst = reinterpret_cast<char *>(scalar_pointer(out + origin));
            W acc = 0;
#pragma unroll
            for (uint32_t z = 0; z < 16; ++z) {
                acc += v[z];
                *reinterpret_cast<W *>(dst + lane_bytes) = rotr1(acc);  // (one offset register for the 16 stores: they share a basic block)
                dst = scalar_pointer(dst + plane_step);
            }
        }
    }
}

}  // namespace wide
}  // namespace ndzip_hip
