// ndzip_amd/csrc/codec_launch.inl -- __global__ kernels and launchers, instantiated for one value type
// (NDZIP_T) per translation unit so the two types compile in parallel.
//
// Kernels (reference counterparts in src/ndzip/cuda_codec.inl, none of whose structure is reused):
//   compress_kernel    persistent, one tile of K hypercubes per workgroup iteration: encode in LDS, publish the
//                      tile length, decoupled look-back for the tile's stream offset, coalesced copy-out, header
//                      entries, stream length.  Replaces compress_block + hierarchical_inclusive_scan (2 kernels x
//                      levels) + compact_all_chunks + store_stream_length (:401-457,:507-511, cuda_bits.cuh:266-333).
//   decompress_kernel  one tile per workgroup, header lookup -> LDS -> decode (replaces decompress_block :477-492).
//   debug_stage_kernel single-hypercube stage entry points for the parity tests.

#include "codec_kernels.hpp"
#include "codec_launch.hpp"

#ifndef NDZIP_T
#error "define NDZIP_T before including codec_launch.inl"
#endif

namespace ndzip_hip {

namespace {

using T_ = NDZIP_T;

// hypercubes per workgroup: f32 pairs two hypercubes (x-adjacent in 3D, so the pair covers whole 128-byte
// lines of the 64-byte cube rows); f64 rows already are >= 128 bytes and the cube is twice as large in LDS.
template<typename T, int Dims>
struct tile_cfg {
    static constexpr int K = sizeof(T) == 4 ? 2 : 1;
    static constexpr int threads = K * threads_per_hc;
    using W = typename word_of<T>::type;
    using L = lds_layout<W>;
    // resident workgroups per CU the LDS admits (160 KiB) -> wavefronts per SIMD the register budget must allow
    static constexpr int min_waves_per_simd = 2;
    static constexpr uint32_t xchg_bytes = 32;  // per hypercube: 2 x uint32 + 2 x W
    static constexpr uint32_t smem_bytes = K * L::cube_bytes + L::zero_bytes + K * xchg_bytes + 32;
};

constexpr unsigned long long st_aggregate = 1ull << 32;
constexpr unsigned long long st_inclusive = 2ull << 32;
constexpr int lookback_loads = 4;  // descriptors per lane and hop: 256 predecessor tiles per hop
constexpr uint32_t spin_limit = 1u << 20;

NDZIP_DEV tile_desc desc_load(const tile_desc *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
NDZIP_DEV void desc_store(tile_desc *p, tile_desc v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

NDZIP_DEV uint32_t wave_sum(uint32_t v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}

// Decoupled look-back, executed by ONE wavefront.  Publishes this tile's aggregate, walks back over the
// predecessors (nearest first, 256 per hop) until an inclusive prefix is found, publishes this tile's
// inclusive prefix and returns its exclusive prefix.  Forward progress: the grid is persistent and fully
// resident and every workgroup handles its tiles in increasing order, so the smallest unfinished tile never
// waits.  Every spin is bounded; on timeout the error word is set and a partial (smaller) sum is returned,
// which keeps all writes inside the caller's buffer.
NDZIP_DEV uint32_t lookback_exclusive_prefix(tile_desc *desc, uint32_t tile, uint32_t aggregate, uint32_t *err, int lane) {
    if (tile == 0) {
        if (lane == 0) desc_store(desc, st_inclusive | aggregate);
        return 0;
    }
    if (lane == 0) desc_store(desc + tile, st_aggregate | aggregate);
    uint32_t exclusive = 0;
    long long base = static_cast<long long>(tile) - 1;
    bool timed_out = false;
    for (;;) {
        tile_desc d[lookback_loads];
        bool found = false;
        int jf = 0, lf = 0;
        for (uint32_t spins = 0;; ++spins) {
#pragma unroll
            for (int j = 0; j < lookback_loads; ++j) {
                const long long idx = base - (j * 64 + lane);
                d[j] = idx >= 0 ? desc_load(desc + idx) : st_inclusive;
            }
            found = false;
            bool stall = false;
#pragma unroll
            for (int j = 0; j < lookback_loads; ++j) {
                const uint32_t status = static_cast<uint32_t>(d[j] >> 32);
                const unsigned long long invalid = __ballot(status == 0);
                const unsigned long long inclusive = __ballot(status == 2);
                if (!found) {
                    if (inclusive != 0) {
                        found = true;
                        jf = j;
                        lf = __builtin_ctzll(inclusive);
                        const unsigned long long nearer = lf == 0 ? 0ull : (~0ull >> (64 - lf));
                        if (invalid & nearer) stall = true;
                    } else if (invalid != 0) {
                        stall = true;
                    }
                }
            }
            if (!stall) break;
            if (spins >= spin_limit) {
                timed_out = true;
                break;
            }
            __builtin_amdgcn_s_sleep(2);
        }
        uint32_t s = 0;
#pragma unroll
        for (int j = 0; j < lookback_loads; ++j) {
            const bool take = !found || j < jf || (j == jf && lane <= lf);
            s += take ? static_cast<uint32_t>(d[j]) : 0u;
        }
        exclusive += wave_sum(s);
        if (found || timed_out) break;
        base -= 64 * lookback_loads;
    }
    if (timed_out && lane == 0) atomicOr(err, 1u);
    if (lane == 0) desc_store(desc + tile, st_inclusive | (exclusive + aggregate));
    return exclusive;
}

template<typename T, int Dims, bool Aligned>
__global__ void __launch_bounds__((tile_cfg<T, Dims>::threads), (tile_cfg<T, Dims>::min_waves_per_simd))
compress_kernel(const typename word_of<T>::type *__restrict__ in, const grid_geom gg, uint32_t *__restrict__ header,
        typename word_of<T>::type *__restrict__ body, tile_desc *desc, uint32_t *out_len, uint32_t len_extra, uint32_t *err) {
    using C = tile_cfg<T, Dims>;
    using W = typename C::W;
    using L = typename C::L;
    constexpr int K = C::K;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = static_cast<int>(threadIdx.x);
    const int grp = tid / threads_per_hc, t = tid % threads_per_hc;
    char *cube = smem + grp * L::cube_bytes;
    char *zero = smem + K * L::cube_bytes;
    uint32_t *xchg = reinterpret_cast<uint32_t *>(zero + L::zero_bytes) + grp * (C::xchg_bytes / 4);
    uint32_t *tile_s = reinterpret_cast<uint32_t *>(zero + L::zero_bytes) + K * (C::xchg_bytes / 4);

    for (uint32_t i = tid; i < L::zero_bytes / 4; i += C::threads) reinterpret_cast<uint32_t *>(zero)[i] = 0;
    // (the first barrier inside forward_transform_hypercube orders this before any stencil read)

    const uint32_t ntiles = (gg.nhc + K - 1) / K;
    for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const uint32_t hc = tile * K + grp;
        const bool active = hc < gg.nhc;
        const uint64_t origin = active ? hc_origin<Dims>(gg, hc) : 0;
        uint32_t len = encode_hypercube<T, Dims, Aligned>(in, gg, origin, active, cube, zero, xchg, t);
        if (!active) len = 0;
        if (t == 0) tile_s[grp] = len;
        __syncthreads();
        if (tid < 64) {
            uint32_t aggregate = 0;
#pragma unroll
            for (int g = 0; g < K; ++g) aggregate += tile_s[g];
            const uint32_t exclusive = lookback_exclusive_prefix(desc, tile, aggregate, err, tid);
            if (tid == 0) tile_s[K] = exclusive;
        }
        __syncthreads();
        uint32_t goff = tile_s[K];
#pragma unroll
        for (int g = 0; g < K; ++g) goff += g < grp ? tile_s[g] : 0u;
        if (active) {
            const W *src = reinterpret_cast<const W *>(cube);
            W *dst = body + goff;
            for (uint32_t w = t; w < len; w += threads_per_hc) dst[w] = src[w];
            if (t == 0) {
                header[hc] = goff + len;  // offset_after(hc), common.hh:342-347
                if (hc == gg.nhc - 1) {
                    if (out_len) *out_len = len_extra + goff + len;
                    // zero the header pad of 64-bit streams with an odd hypercube count (cuda_codec.inl:446-452)
                    if (sizeof(W) == 8 && (gg.nhc & 1u)) header[gg.nhc] = 0;
                }
            }
        }
        __syncthreads();  // copy-out done before the next tile overwrites the staging buffers / tile_s
    }
}

template<typename T, int Dims, bool Aligned>
__global__ void __launch_bounds__((tile_cfg<T, Dims>::threads))
decompress_kernel(const uint32_t *__restrict__ header, const uint32_t *__restrict__ header_base_ptr, const typename word_of<T>::type *__restrict__ body,
        typename word_of<T>::type *__restrict__ out, const grid_geom gg, uint32_t *err) {
    using C = tile_cfg<T, Dims>;
    using W = typename C::W;
    using L = typename C::L;
    using P = profile<T, Dims>;
    constexpr int K = C::K;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = static_cast<int>(threadIdx.x);
    const int grp = tid / threads_per_hc, t = tid % threads_per_hc;
    char *cube = smem + grp * L::cube_bytes;
    uint32_t *xchg = reinterpret_cast<uint32_t *>(smem + K * L::cube_bytes + L::zero_bytes) + grp * (C::xchg_bytes / 4);

    const uint32_t hc = blockIdx.x * K + grp;
    const bool active = hc < gg.nhc;
    uint32_t begin = 0, len = 0;
    if (active) {
        const uint32_t header_base = header_base_ptr ? *header_base_ptr : 0u;
        begin = (hc ? header[hc - 1] : header_base) - header_base;  // stream<Profile>::hypercube, common.hh:350-358
        len = header[hc] - header_base - begin;
        if (len < static_cast<uint32_t>(P::head_words) || len > static_cast<uint32_t>(P::max_hc_words)) {
            if (t == 0) atomicOr(err, 2u);  // corrupt header
            len = 0;
        }
    }
    W *dst = reinterpret_cast<W *>(cube);
    if (len == 0) {
        // padding group or corrupt entry: decode an all-zero hypercube so every LDS index stays in range
        if (t < P::head_words) dst[t] = 0;
    } else {
        const W *src = body + begin;
        for (uint32_t w = t; w < len; w += threads_per_hc) dst[w] = src[w];
    }
    __syncthreads();
    decode_hypercube<T, Dims, Aligned>(out, gg, active ? hc_origin<Dims>(gg, hc) : 0, active && len != 0, cube, xchg, t);
}

// ---- stage kernels for the parity tests: exactly one hypercube, 128 work-items ---------------------------------

template<typename T, int Dims, bool Aligned>
__global__ void __launch_bounds__(threads_per_hc)
debug_stage_kernel(int stage, const grid_geom gg, uint32_t hc, const typename word_of<T>::type *__restrict__ in,
        typename word_of<T>::type *__restrict__ out, uint32_t *out_len) {
    using W = typename word_of<T>::type;
    using L = lds_layout<W>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *cube = smem;
    char *zero = smem + L::cube_bytes;
    uint32_t *xchg = reinterpret_cast<uint32_t *>(zero + L::zero_bytes);
    const int t = static_cast<int>(threadIdx.x);
    for (uint32_t i = t; i < L::zero_bytes / 4; i += threads_per_hc) reinterpret_cast<uint32_t *>(zero)[i] = 0;

    W r[vals_per_thread];
    if (stage == debug_forward_transform) {
        forward_transform_hypercube<T, Dims, Aligned>(in, gg, hc_origin<Dims>(gg, hc), true, cube, zero, t, r);
        for (int j = 0; j < vals_per_thread; ++j) out[t * 32 + j] = r[j];
    } else if (stage == debug_encode_residuals) {
        for (int j = 0; j < vals_per_thread; ++j) r[j] = in[t * 32 + j];
        __syncthreads();
        const uint32_t len = encode_residuals<T, Dims>(r, cube, xchg, t);
        const W *src = reinterpret_cast<const W *>(cube);
        for (uint32_t w = t; w < len; w += threads_per_hc) out[w] = src[w];
        if (t == 0) *out_len = len;
    } else if (stage == debug_decode_residuals) {
        W *dst = reinterpret_cast<W *>(cube);
        for (uint32_t w = t; w < profile<T, Dims>::max_hc_words; w += threads_per_hc) dst[w] = in[w];
        __syncthreads();
        decode_residuals<T, Dims>(cube, xchg, t, r);
        for (int j = 0; j < vals_per_thread; ++j) out[t * 32 + j] = r[j];
    } else if (stage == debug_inverse_transform) {
        for (int j = 0; j < vals_per_thread; ++j) r[j] = in[t * 32 + j];
        __syncthreads();
        inverse_transform_hypercube<T, Dims, Aligned>(r, out, gg, hc_origin<Dims>(gg, hc), true, cube, xchg, t);
    }
}

__global__ void debug_transpose_kernel(const uint32_t *in, uint32_t *out, uint32_t n, int generic) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t x[32];
#pragma unroll
    for (int j = 0; j < 32; ++j) x[j] = in[i * 32 + j];
    if (generic) {
        transpose32_generic(x);
    } else {
        transpose32(x);
    }
#pragma unroll
    for (int j = 0; j < 32; ++j) out[i * 32 + j] = x[j];
}

template<typename T, int Dims, bool Aligned>
hipError_t launch_compress_profile(const compress_args &a) {
    using C = tile_cfg<T, Dims>;
    using W = typename C::W;
    const uint32_t ntiles = (a.gg.nhc + C::K - 1) / C::K;
    if (ntiles == 0) return hipSuccess;
    auto kernel = compress_kernel<T, Dims, Aligned>;
    // persistent grid, fully resident: bounded by the occupancy query and by what the LDS alone admits
    static int blocks_per_cu = 0;
    if (blocks_per_cu == 0) {
        int api = 0;
        hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&api, kernel, C::threads, C::smem_bytes);
        if (e != hipSuccess) return e;
        const int by_lds = static_cast<int>((160u * 1024u) / C::smem_bytes);
        blocks_per_cu = api < by_lds ? api : by_lds;
        if (blocks_per_cu < 1) blocks_per_cu = 1;
    }
    uint32_t grid = static_cast<uint32_t>(a.num_cus) * static_cast<uint32_t>(blocks_per_cu);
    if (grid > ntiles) grid = ntiles;
    hipError_t e = hipMemsetAsync(a.desc, 0, static_cast<size_t>(ntiles) * sizeof(tile_desc), a.stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(C::threads), C::smem_bytes, a.stream, static_cast<const W *>(a.in), a.gg,
            a.header, static_cast<W *>(a.body), a.desc, a.out_len, a.len_extra, a.err);
    return hipGetLastError();
}

template<typename T, int Dims, bool Aligned>
hipError_t launch_decompress_profile(const decompress_args &a) {
    using C = tile_cfg<T, Dims>;
    using W = typename C::W;
    const uint32_t ntiles = (a.gg.nhc + C::K - 1) / C::K;
    if (ntiles == 0) return hipSuccess;
    hipLaunchKernelGGL((decompress_kernel<T, Dims, Aligned>), dim3(ntiles), dim3(C::threads), C::smem_bytes, a.stream,
            a.header, a.header_base, static_cast<const W *>(a.body), static_cast<W *>(a.out), a.gg, a.err);
    return hipGetLastError();
}

template<typename T, int Dims, bool Aligned>
hipError_t launch_debug_profile(int stage, const grid_geom &gg, uint32_t hc, const void *in, void *out, uint32_t *out_len,
        hipStream_t stream) {
    using W = typename word_of<T>::type;
    using L = lds_layout<W>;
    const uint32_t smem = L::cube_bytes + L::zero_bytes + 64;
    hipLaunchKernelGGL((debug_stage_kernel<T, Dims, Aligned>), dim3(1), dim3(threads_per_hc), smem, stream, stage, gg, hc,
            static_cast<const W *>(in), static_cast<W *>(out), out_len);
    return hipGetLastError();
}

#define NDZIP_DISPATCH(FN, dims, aligned, ...)                                          \
    switch (dims) {                                                                     \
        case 1: return (aligned) ? FN<T_, 1, true>(__VA_ARGS__) : FN<T_, 1, false>(__VA_ARGS__); \
        case 2: return (aligned) ? FN<T_, 2, true>(__VA_ARGS__) : FN<T_, 2, false>(__VA_ARGS__); \
        case 3: return (aligned) ? FN<T_, 3, true>(__VA_ARGS__) : FN<T_, 3, false>(__VA_ARGS__); \
        default: return hipErrorInvalidValue;                                           \
    }

}  // namespace

template<>
int compress_hcs_per_group<T_>(int) {
    return tile_cfg<T_, 1>::K;
}

template<>
uint32_t compress_num_tiles<T_>(int, uint32_t nhc) {
    return (nhc + tile_cfg<T_, 1>::K - 1) / tile_cfg<T_, 1>::K;
}

template<>
hipError_t launch_compress<T_>(int dims, const compress_args &a) {
    NDZIP_DISPATCH(launch_compress_profile, dims, a.aligned, a)
}

template<>
hipError_t launch_decompress<T_>(int dims, const decompress_args &a) {
    NDZIP_DISPATCH(launch_decompress_profile, dims, a.aligned, a)
}

template<>
hipError_t launch_debug_stage<T_>(int stage, int dims, const grid_geom &gg, uint32_t hc, const void *in, void *out,
        uint32_t *out_len, uint32_t n, bool aligned, hipStream_t stream) {
    if (stage == debug_transpose32 || stage == debug_transpose32_generic) {
        if (n == 0) return hipSuccess;
        hipLaunchKernelGGL(debug_transpose_kernel, dim3((n + 63) / 64), dim3(64), 0, stream, static_cast<const uint32_t *>(in),
                static_cast<uint32_t *>(out), n, stage == debug_transpose32_generic ? 1 : 0);
        return hipGetLastError();
    }
    NDZIP_DISPATCH(launch_debug_profile, dims, aligned, stage, gg, hc, in, out, out_len, stream)
}

#undef NDZIP_DISPATCH

}  // namespace ndzip_hip
