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
