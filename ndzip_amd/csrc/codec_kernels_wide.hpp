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
        const bool odd = (t & 1) != 0;
#pragma unroll
        for (int j = 0; j < vals; ++j) {
            const uint32_t hi = static_cast<uint32_t>(r[j] >> 32), lo = static_cast<uint32_t>(r[j]);
            const uint32_t got = pair_swap(odd ? hi : lo);  // the even lane sends its low dwords, the odd lane its high ones
            planes[j] = odd ? got : hi;                      // values 0..15 of the pair: the even lane's
            planes[vals + j] = odd ? lo : got;               // values 16..31 of the pair: the odd lane's
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
        uint32_t a = lds_address(run32 + h.slot);  // (a running linear LDS address: one increment per kept plane)
#pragma unroll
        for (int i = 0; i < planes_per_lane; ++i) {
            if ((h.head_bits >> (31 - i)) & 1u) {
                *reinterpret_cast<uint32_t *>(lds_pointer(R::at(a))) = planes[i];
                a += 8;
            }
        }
    }
};

}  // namespace wide
}  // namespace ndzip_hip
