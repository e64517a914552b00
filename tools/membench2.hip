// tools/membench2.hip -- from the pattern-faithful streaming loop (tools/membench.hip) towards the compress kernel's
// skeleton, one feature at a time, to find what halves the load rate:
//   bit 0  register prefetch one tile ahead (else load at the top of the iteration)
//   bit 1  dynamic tickets (16 classes, synchronous atomic at the top)
//   bit 2  two extra workgroup barriers per iteration
//   bit 3  prefetch split 2 + 6 vectors around a barrier
//   bit 4  store 0.687 N back (tile-contiguous)
//   bit 5  tickets drawn one iteration ahead
#include <hip/hip_runtime.h>
#include <cstdio>
#include "codec_common.hpp"
#include "codec_kernels.hpp"

using namespace ndzip_hip;

__global__ void __launch_bounds__(256, 3) skel(const uint32_t *in, grid_geom gg, uint32_t *out, uint32_t out_words, uint32_t *tickets_base, uint32_t f, uint32_t ncls, uint32_t tstride) {
    using L = lds_layout<uint32_t>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ uint32_t s_ticket;
    const int tid = threadIdx.x, grp = tid / 128, t = tid % 128;
    char *cube = smem + grp * L::cube_bytes;
    const uint32_t ntiles = gg.nhc / 2;
    const uint32_t cls = blockIdx.x % ncls;
    uint32_t *tickets = tickets_base + cls * tstride - cls;  // counter of class c at word c * tstride
    uint32_t ahead = 0;
    uint32_t tile = blockIdx.x;
    if (f & 2u) {
        if (tid == 0) {
            s_ticket = atomicAdd(tickets + cls, 1u);
            if (f & 32u) ahead = atomicAdd(tickets + cls, 1u);
        }
        __syncthreads();
        tile = s_ticket * ncls + cls;
    }
    input_regs<uint32_t, true> pre;
    if ((f & 1u) && tile < ntiles) load_hypercube_regs<float, 3, true>(in, gg, hc_origin<3>(gg, tile * 2 + grp), t, pre);
    uint32_t sink = 0;
    while (tile < ntiles) {
        if (!(f & 1u)) load_hypercube_regs<float, 3, true>(in, gg, hc_origin<3>(gg, tile * 2 + grp), t, pre);
        uint32_t next = tile + gridDim.x;
        if (f & 2u) {
            if (f & 32u) {
                stage_hypercube_regs<uint32_t, true>(pre, cube, t);
                if (tid == 0) {
                    s_ticket = ahead;
                    ahead = atomicAdd(tickets + cls, 1u);
                }
            } else {
                uint32_t nt = 0;
                if (tid == 0) nt = atomicAdd(tickets + cls, 1u);
                stage_hypercube_regs<uint32_t, true>(pre, cube, t);
                if (tid == 0) s_ticket = nt;
            }
        } else {
            stage_hypercube_regs<uint32_t, true>(pre, cube, t);
        }
        __syncthreads();
        if (f & 2u) next = s_ticket * ncls + cls;
        uint32_t nclamp = next < ntiles ? next : ntiles - 1;
        const uint64_t norigin = hc_origin<3>(gg, nclamp * 2 + grp);
        if (f & 1u) {
            if (f & 8u) {
                load_hypercube_regs<float, 3, true, 0, 2>(in, gg, norigin, t, pre);
                sink += reinterpret_cast<const uint32_t *>(smem)[(tid * 33) % 8192];
                __syncthreads();
                load_hypercube_regs<float, 3, true, 1, 2>(in, gg, norigin, t, pre);
            } else {
                load_hypercube_regs<float, 3, true>(in, gg, norigin, t, pre);
            }
        }
        if (f & 4u) {
            sink += reinterpret_cast<const uint32_t *>(smem)[(tid * 35) % 8192];
            __syncthreads();
        }
        if (f & 16u) {
            const uint32_t *src = reinterpret_cast<const uint32_t *>(smem);
            vec16 *dst = reinterpret_cast<vec16 *>(out + static_cast<size_t>(tile) * out_words);
            for (uint32_t v = tid; v < out_words / 4; v += 256) dst[v] = *reinterpret_cast<const vec16 *>(src + 4 * v);
        } else {
            sink += reinterpret_cast<const uint32_t *>(smem)[(tid * 33) % 8192];
        }
        __syncthreads();
        tile = next;
    }
    if (sink == 0x12345678) out[0] = sink;
}

int main() {
    const uint32_t ext[3] = {512, 512, 512};
    const grid_geom gg = make_geom(3, ext);
    const size_t n = 512ull * 512 * 512;
    uint32_t *in, *out, *tickets;
    hipMalloc(&in, n * 4);
    hipMalloc(&out, n * 4 + (1 << 20));
    hipMalloc(&tickets, 1 << 20);
    hipMemset(in, 1, n * 4);
    const uint32_t out_words = 5624;
    const uint32_t smem = 2 * lds_layout<uint32_t>::cube_bytes;
    struct cfg { uint32_t f, ncls, stride; };
    const cfg cfgs[] = {{1, 16, 1}, {3, 16, 1}, {3, 16, 16}, {3, 16, 32}, {3, 16, 64}, {3, 16, 1024}, {3, 32, 64}, {3, 64, 64}, {3, 64, 1024}, {3, 256, 64}, {3, 768, 64},
                        {19, 16, 1}, {19, 16, 64}, {19, 64, 64}, {19, 64, 1024}, {17, 16, 1}, {31, 64, 64}, {63, 64, 64}};
    for (const cfg &c : cfgs) {
        const uint32_t f = c.f;
        hipEvent_t a, b;
        hipEventCreate(&a);
        hipEventCreate(&b);
        float best = 1e9f, sum = 0;
        for (int rep = 0; rep < 8; ++rep) {
            hipMemsetAsync(tickets, 0, 1 << 20);
            hipEventRecord(a);
            hipLaunchKernelGGL(skel, dim3(768), dim3(256), smem, 0, in, gg, out, out_words, tickets, f, c.ncls, c.stride);
            hipEventRecord(b);
            hipEventSynchronize(b);
            float ms;
            hipEventElapsedTime(&ms, a, b);
            if (rep > 1) { sum += ms; if (ms < best) best = ms; }
        }
        printf("flags %2u classes %3u stride %4u words [%s%s%s%s%s%s]: best %.3f ms avg %.3f ms\n", f, c.ncls, c.stride, f & 1 ? "prefetch " : "", f & 2 ? "tickets " : "", f & 4 ? "barriers " : "",
               f & 8 ? "split " : "", f & 16 ? "store " : "", f & 32 ? "ahead " : "", best, sum / 6);
    }
    return 0;
}
