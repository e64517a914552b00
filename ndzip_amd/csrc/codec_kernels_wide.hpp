// ndzip_amd/csrc/codec_kernels_wide.hpp -- f64 ENCODE with 256 work-items per hypercube ("wide" mapping).
//
// The 128-work-item mapping of codec_kernels.hpp gives an f64 work-item 32 values = 64 VGPRs of residuals and 64 VGPRs
// of plane words: too much to keep an encoded tile in registers across an iteration, so the f64 compress kernel had to
// write a tile out in the iteration that encoded it, with its look-back on the critical path.  Here a work-item owns 16
// consecutive cube-local values; a 64-value chunk spans 4 lanes.  Per work-item that is 32 VGPRs of values, 32 VGPRs of
// prefetch and -- after the transpose -- 32 VGPRs of plane words: exactly the register picture of the f32 kernel, so the
// register-buffered deferred-write-out pipeline (compress_kernel_db) carries over.
//
// Bit transpose of a chunk (64 values x 64 bits) over its lane quad q = 0..3 (values 16q .. 16q+15):
//   1. the two lanes of a pair (0,1) / (2,3) swap halves: the even lane ends up with the HIGH dwords of the pair's 32
//      values, the odd lane with the LOW dwords (16 DPP moves each);
//   2. one 32x32 transpose per lane (the f32 network).
// Lane 0 then holds, for the planes 0..31 (bits 63..32), the dword that covers values 0..31 = the HIGH dword of the
// 64-bit plane word; lane 2 the LOW dword of the same planes (values 32..63); lanes 1 / 3 the same for planes 32..63.
// Each lane compacts its 32 dwords (32 conditional writes instead of the narrow mapping's 64).
#pragma once

#include "codec_kernels.hpp"

namespace ndzip_hip {
namespace wide {

constexpr int threads = 256;  // work-items per hypercube
constexpr int vals = 16;      // cube-local values [16 t, 16 t + 16) per work-item
using W = uint64_t;
constexpr uint32_t head_words = hc_size / 64;  // 64 chunk heads

struct layout {
    static constexpr uint32_t chunk_bytes = vals * sizeof(W) + 16;      // 144: the f32 mapping's lane stride
    static constexpr uint32_t cube_bytes = threads * chunk_bytes;        // 36864
    static constexpr uint32_t zero_bytes = vals * sizeof(W) + 256;
    // slot (mod 16) no in-cube lane of a 16-lane group touches in the neighbour reads: lane t sits in slot 9t mod 16;
    // rows y-1 (lane t-1, 3D), (z-1,y-1) (t-17) and the 2D row above (t-4) leave slot 7 free where the border lanes are
    static constexpr uint32_t zero_offset = 7 * 16;
    NDZIP_DEV static constexpr uint32_t off(uint32_t k) { return k * 8u + (k >> 4) * 16u; }
};

struct input_regs {
    static constexpr int NV = hc_size / 2 / threads;  // 8 vectors of two values
    vec16 v[NV];
};

// coalesced global loads: vector i of work-item t covers values (i*256 + t)*2; 512 values are whole rows / planes
template<int Dims, bool Aligned, int Part = -1, int Split = 0>
NDZIP_DEV void load_regs(const W *__restrict__ in, const grid_geom &gg, uint64_t origin, int t, input_regs &regs) {
    constexpr int first = Part == 1 ? Split : 0;
    constexpr int last = Part == 0 ? Split : input_regs::NV;
    const W *base = in + origin + local_offset<Dims>(gg, static_cast<uint32_t>(t) * 2u);
    const uint64_t step = local_offset<Dims>(gg, threads * 2);
#pragma unroll
    for (int i = first; i < last; ++i) regs.v[i] = global_load16<Aligned>(base + i * step);
}

NDZIP_DEV void stage_regs(const input_regs &regs, char *cube, int t) {
    char *base = cube + layout::off(static_cast<uint32_t>(t) * 2u);
    constexpr uint32_t step = layout::off(threads * 2);
#pragma unroll
    for (int i = 0; i < input_regs::NV; ++i) {
        vec16 r;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const uint64_t x = rotl1(static_cast<uint64_t>(regs.v[i].w[2 * j]) | (static_cast<uint64_t>(regs.v[i].w[2 * j + 1]) << 32));
            r.w[2 * j] = static_cast<uint32_t>(x);
            r.w[2 * j + 1] = static_cast<uint32_t>(x >> 32);
        }
        lds_write16(base + i * step, r);
    }
}

// fused Lorenzo stencil out of LDS + complement_negative -> residuals r[16] of work-item t
template<int Dims>
NDZIP_DEV void stencil(const char *cube, const char *zero, int t, W (&r)[vals]) {
    const uint32_t k0 = static_cast<uint32_t>(t) * vals;
    const char *own = cube + layout::off(k0);
    constexpr int Q = 4;  // values folded at a time (two 16-byte reads per row)
    if constexpr (Dims == 1) {
        W prev = lds_read<W>(t > 0 ? cube + layout::off(k0 - 1) : zero, 0);
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
        const char *up = y > 0 ? cube + layout::off(k0 - 64) : zero;
        const W ol = lds_read<W>(qx ? cube + layout::off(k0 - 1) : zero, 0);
        const W ul = lds_read<W>((qx && y > 0) ? cube + layout::off(k0 - 65) : zero, 0);
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
        const char *row_p = y > 0 ? cube + layout::off(k0 - 16) : zero;                  // (z, y-1)
        const char *row_a1 = z > 0 ? cube + layout::off(k0 - 256) : zero;               // (z-1, y)
        const char *row_p1 = (z > 0 && y > 0) ? cube + layout::off(k0 - 256 - 16) : zero;  // (z-1, y-1)
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

// chunk head (64 bits, same on the 4 lanes of the chunk) of the residuals of work-item t
NDZIP_DEV void chunk_head(const W (&r)[vals], uint32_t &head_hi, uint32_t &head_lo) {
    W own = 0;
#pragma unroll
    for (int j = 0; j < vals; ++j) own |= r[j];
    head_hi = quad_or(static_cast<uint32_t>(own >> 32));
    head_lo = quad_or(static_cast<uint32_t>(own));
}

// residuals -> this lane's 32 plane dwords (see the file comment); q = t & 3
NDZIP_DEV void transpose_chunk(const W (&r)[vals], int t, uint32_t (&planes)[32]) {
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

}  // namespace wide
}  // namespace ndzip_hip
