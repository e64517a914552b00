// tools/membench.hip -- memory-path ceiling of the compress kernel's access pattern (experiments only).
//
// Same global addressing as compress_kernel<float,3>: persistent workgroups of 256 work-items, each iteration
// loads two x-adjacent 16^3 hypercubes of a 512^3 float grid with the codec's own load/stage helpers, then
// writes `out_words` words per tile contiguously (tile * out_words) from LDS.  No stencil, no transposes, no
// look-back: what is left is HBM + LDS staging + barriers.  Modes: 0 = load only, 1 = load + store,
// 2 = load (prefetched one iteration ahead) + store.
//
// build+run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -std=c++17 -I ndzip_amd/csrc tools/membench.hip -o /tmp/membench && /tmp/membench
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#include "codec_kernels.hpp"

using namespace ndzip_hip;

// Mode 4 of the load test: the 256 work-items load the x-adjacent hypercube PAIR as 128-byte row segments (8 lanes x
// 16 B per row: 4 lanes in the first cube, 4 in the second), 32 rows per wave-instruction-pair... i.e. every vector
// load touches whole 128-byte lines instead of two half lines.
__global__ void __launch_bounds__(256) paired_load_kernel(const uint32_t *in, grid_geom gg, uint32_t *out) {
    using L = lds_layout<uint32_t>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const uint32_t ntiles = gg.nhc / 2;
    uint32_t sink = 0;
    for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const uint64_t origin = hc_origin<3>(gg, tile * 2);  // first cube of the pair; the second is 16 values further in x
        vec16 v[8];
        // 512 rows of 16+16 values; work-item handles row (i*32 + tid/8), 16-byte piece tid%8 of the 128-byte pair row
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const uint32_t row = i * 32 + tid / 8;           // 0..255 = z*16 + y
            const uint32_t piece = tid % 8;                   // 0..3 -> cube 0, 4..7 -> cube 1
            const uint64_t off = static_cast<uint64_t>(row >> 4) * gg.stride[0] + static_cast<uint64_t>(row & 15) * gg.stride[1] + piece * 4;
            v[i] = *reinterpret_cast<const vec16 *>(in + origin + off);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const uint32_t row = i * 32 + tid / 8;
            const uint32_t piece = tid % 8;
            const uint32_t k = row * 16 + (piece & 3) * 4;
            lds_write16(smem + (piece >> 2) * L::cube_bytes + L::off(k), v[i]);
        }
        __syncthreads();
        sink += reinterpret_cast<const uint32_t *>(smem)[(tid * 33) % 8192];
        __syncthreads();
    }
    if (sink == 0x12345678) out[0] = sink;
}

template<int Mode>
__global__ void __launch_bounds__(256) pattern_kernel(const uint32_t *in, grid_geom gg, uint32_t *out, uint32_t out_words, uint32_t busy) {
    using L = lds_layout<uint32_t>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, grp = tid / 128, t = tid % 128;
    char *cube = smem + grp * L::cube_bytes;
    const uint32_t ntiles = gg.nhc / 2;
    input_regs<uint32_t, true> pre;
    uint32_t tile = blockIdx.x;
    if (Mode == 2 && tile < ntiles) load_hypercube_regs<float, 3, true>(in, gg, hc_origin<3>(gg, tile * 2 + grp), t, pre);
    uint32_t sink = 0;
    for (; tile < ntiles; tile += gridDim.x) {
        if (Mode != 2) load_hypercube_regs<float, 3, true>(in, gg, hc_origin<3>(gg, tile * 2 + grp), t, pre);
        stage_hypercube_regs<uint32_t, true>(pre, cube, t);
        if (Mode == 2) {
            uint32_t next = tile + gridDim.x;
            if (next >= ntiles) next = ntiles - 1;
            load_hypercube_regs<float, 3, true>(in, gg, hc_origin<3>(gg, next * 2 + grp), t, pre);
        }
        __syncthreads();
        // emulated compute phase: `busy` dependent VALU ops per work-item (keeps the SIMD issuing, no memory traffic)
        {
            uint32_t x = tid;
            for (uint32_t i = 0; i < busy; ++i) x = x * 1664525u + 1013904223u;
            asm volatile("" ::"v"(x));
        }
        if (Mode == 0) {
            sink += reinterpret_cast<const uint32_t *>(smem)[(tid * 33) % 8192];
        } else {
            const uint32_t *src = reinterpret_cast<const uint32_t *>(smem);
            vec16 *dst = reinterpret_cast<vec16 *>(out + static_cast<size_t>(tile) * out_words);
            for (uint32_t v = tid; v < out_words / 4; v += 256) {
                vec16 x;
                for (int j = 0; j < 4; ++j) x.w[j] = src[4 * v + j];
                dst[v] = x;
            }
        }
        __syncthreads();
    }
    if (Mode == 0 && sink == 0x12345678) out[0] = sink;
}

// plain streaming kernels for calibration: read n vec16, write m vec16 (m <= n), grid-stride, no LDS
__global__ void __launch_bounds__(256) plain_kernel(const vec16 *in, vec16 *out, size_t n, size_t m) {
    const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
    uint32_t acc = 0;
    for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
        const vec16 v = in[i];
        if (i < m) {
            out[i] = v;
        } else {
            acc += v.w[0] ^ v.w[3];
        }
    }
    if (acc == 0x12345678) out[0].w[0] = acc;
}

float run_plain(const uint32_t *in, uint32_t *out, size_t n_bytes, size_t m_bytes, int blocks) {
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    float best = 1e9f;
    for (int rep = 0; rep < 6; ++rep) {
        hipEventRecord(a);
        hipLaunchKernelGGL(plain_kernel, dim3(blocks), dim3(256), 0, 0, reinterpret_cast<const vec16 *>(in), reinterpret_cast<vec16 *>(out),
                n_bytes / 16, m_bytes / 16);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        if (rep > 0 && ms < best) best = ms;
    }
    return best;
}

template<int Mode>
float run(const uint32_t *in, const grid_geom &gg, uint32_t *out, uint32_t out_words, int blocks_per_cu, uint32_t busy = 0) {
    const uint32_t smem = 2 * lds_layout<uint32_t>::cube_bytes;
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    float best = 1e9f;
    for (int rep = 0; rep < 6; ++rep) {
        hipEventRecord(a);
        hipLaunchKernelGGL(pattern_kernel<Mode>, dim3(256 * blocks_per_cu), dim3(256), smem, 0, in, gg, out, out_words, busy);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        if (rep > 0 && ms < best) best = ms;
    }
    return best;
}

int main() {
    const uint32_t ext[3] = {512, 512, 512};
    const grid_geom gg = make_geom(3, ext);
    const size_t n = 512ull * 512 * 512;
    uint32_t *in, *out;
    hipMalloc(&in, n * 4);
    hipMalloc(&out, n * 4 + (1 << 20));
    hipMemset(in, 1, n * 4);
    const uint32_t out_words = 5624;  // 2 x 2812 words per tile = ratio 0.6866
    for (int blocks : {2048, 8192, 32768}) {
        const float r = run_plain(in, out, n * 4, 0, blocks);
        const float c = run_plain(in, out, n * 4, n * 4, blocks);
        const float mix = run_plain(in, out, n * 4, static_cast<size_t>(16384) * out_words * 4, blocks);
        printf("plain grid-stride, %5d blocks: read only %.3f ms (%.0f GB/s) | copy %.3f ms (%.0f GB/s) | read N + write 0.687N %.3f ms (%.0f GB/s)\n", blocks,
                r, n * 4 / r / 1e6, c, 2.0 * n * 4 / c / 1e6, mix, (n * 4 + 16384.0 * out_words * 4) / mix / 1e6);
    }
    for (int bpc = 2; bpc <= 4; ++bpc) {
        hipEvent_t a, b;
        hipEventCreate(&a);
        hipEventCreate(&b);
        float best = 1e9f;
        for (int rep = 0; rep < 6; ++rep) {
            hipEventRecord(a);
            hipLaunchKernelGGL(paired_load_kernel, dim3(256 * bpc), dim3(256), 2 * lds_layout<uint32_t>::cube_bytes, 0, in, gg, out);
            hipEventRecord(b);
            hipEventSynchronize(b);
            float ms;
            hipEventElapsedTime(&ms, a, b);
            if (rep > 0 && ms < best) best = ms;
        }
        printf("blocks/CU %d: paired 128-byte-row load only %.3f ms (%.0f GB/s)\n", bpc, best, n * 4 / best / 1e6);
    }
    for (int bpc = 1; bpc <= 4; ++bpc) {
        const float t0 = run<0>(in, gg, out, out_words, bpc);
        const float t1 = run<1>(in, gg, out, out_words, bpc);
        const float t2 = run<2>(in, gg, out, out_words, bpc);
        printf("blocks/CU %d: load only %.3f ms (%.0f GB/s) | load+store %.3f ms (%.0f GB/s) | prefetch+store %.3f ms (%.0f GB/s)\n", bpc,
                t0, n * 4 / t0 / 1e6, t1, (n * 4 + 16384.0 * out_words * 4) / t1 / 1e6, t2, (n * 4 + 16384.0 * out_words * 4) / t2 / 1e6);
    }
    // emulated compute: `busy` dependent multiply-adds per work-item between load and store
    for (uint32_t busy : {0u, 200u, 400u, 800u, 1600u}) {
        for (int bpc = 2; bpc <= 4; ++bpc) {
            printf("busy %4u blocks/CU %d: load+store %.3f ms | prefetch+store %.3f ms\n", busy, bpc, run<1>(in, gg, out, out_words, bpc, busy),
                    run<2>(in, gg, out, out_words, bpc, busy));
        }
    }
    return 0;
}
