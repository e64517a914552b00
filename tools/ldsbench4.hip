// tools/ldsbench4.hip -- LDS read THROUGHPUT by wall clock (the SQ_LDS_IDX_ACTIVE counter under-reports ds_read_b128):
// every CU runs 8 wavefronts that each issue a long stream of reads of one instruction kind with the stencil's lane
// addressing (stride = chunk bytes).  Prints ns per wave-instruction per CU and bytes/ns/CU.
#include <hip/hip_runtime.h>
#include <cstdio>

template<int Op>
__global__ void __launch_bounds__(512) k(uint32_t *out, uint32_t stride, int iters, int zero_mode) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (uint32_t i = tid; i < 40000 / 4; i += 512) reinterpret_cast<uint32_t *>(smem)[i] = i;
    __syncthreads();
    const uint32_t base = static_cast<uint32_t>(reinterpret_cast<uintptr_t>((const __attribute__((address_space(3))) char *) smem));
    const uint32_t t = (wave & 1) * 64 + lane;
    uint32_t addr = base + t * stride;
    if (zero_mode == 1 && (t & 7) == 0) addr = base + 128 * stride;            // zero block in slot 0
    if (zero_mode == 2 && (t & 7) == 0) addr = base + 128 * stride + 48;       // zero block in slot 3
    uint32_t acc = 0;
    for (int it = 0; it < iters; ++it) {
        uint4 v0, v1, v2, v3;
        if (Op == 0) {
            asm volatile("ds_read_b128 %0, %4\n ds_read_b128 %1, %4 offset:16\n ds_read_b128 %2, %4 offset:32\n ds_read_b128 %3, %4 offset:48\n s_waitcnt lgkmcnt(0)"
                         : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3) : "v"(addr) : "memory");
        } else if (Op == 1) {
            asm volatile("ds_read2_b64 %0, %4 offset0:0 offset1:1\n ds_read2_b64 %1, %4 offset0:2 offset1:3\n ds_read2_b64 %2, %4 offset0:4 offset1:5\n ds_read2_b64 %3, %4 offset0:6 offset1:7\n s_waitcnt lgkmcnt(0)"
                         : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3) : "v"(addr) : "memory");
        } else if (Op == 2) {
            uint2 a, b, c, d, e, f, g, h;
            asm volatile("ds_read_b64 %0, %8\n ds_read_b64 %1, %8 offset:8\n ds_read_b64 %2, %8 offset:16\n ds_read_b64 %3, %8 offset:24\n"
                         "ds_read_b64 %4, %8 offset:32\n ds_read_b64 %5, %8 offset:40\n ds_read_b64 %6, %8 offset:48\n ds_read_b64 %7, %8 offset:56\n s_waitcnt lgkmcnt(0)"
                         : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d), "=&v"(e), "=&v"(f), "=&v"(g), "=&v"(h) : "v"(addr) : "memory");
            v0 = {a.x, a.y, b.x, b.y}; v1 = {c.x, c.y, d.x, d.y}; v2 = {e.x, e.y, f.x, f.y}; v3 = {g.x, g.y, h.x, h.y};
        } else if (Op == 3) {
            uint2 a, b, c, d, e, f, g, h;
            asm volatile("ds_read2_b32 %0, %8 offset0:0 offset1:1\n ds_read2_b32 %1, %8 offset0:2 offset1:3\n ds_read2_b32 %2, %8 offset0:4 offset1:5\n ds_read2_b32 %3, %8 offset0:6 offset1:7\n"
                         "ds_read2_b32 %4, %8 offset0:8 offset1:9\n ds_read2_b32 %5, %8 offset0:10 offset1:11\n ds_read2_b32 %6, %8 offset0:12 offset1:13\n ds_read2_b32 %7, %8 offset0:14 offset1:15\n s_waitcnt lgkmcnt(0)"
                         : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d), "=&v"(e), "=&v"(f), "=&v"(g), "=&v"(h) : "v"(addr) : "memory");
            v0 = {a.x, a.y, b.x, b.y}; v1 = {c.x, c.y, d.x, d.y}; v2 = {e.x, e.y, f.x, f.y}; v3 = {g.x, g.y, h.x, h.y};
        } else {
            // ds_read_b96 + b32?  not useful; Op 4 = b128 contiguous reference handled by stride 16
        }
        acc += v0.x + v1.y + v2.z + v3.w;
    }
    out[blockIdx.x * 512 + tid] = acc;
}

template<int Op>
void run(const char *name, uint32_t *out, uint32_t stride, int zero_mode) {
    const int iters = 4096, grid = 256;
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    hipLaunchKernelGGL(k<Op>, dim3(grid), dim3(512), 65536, 0, out, stride, iters, zero_mode);
    hipEventRecord(a);
    hipLaunchKernelGGL(k<Op>, dim3(grid), dim3(512), 65536, 0, out, stride, iters, zero_mode);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    const double bytes_per_cu = 8.0 * 64 * 64 * iters;  // 8 waves x 64 lanes x 64 bytes per iteration
    printf("%-12s stride %3u zero %d: %.3f ms  %.1f B/ns/CU  (%.2f ns per 1 KiB wave-read)\n", name, stride, zero_mode, ms,
           bytes_per_cu / (ms * 1e6), ms * 1e6 / (8.0 * 4 * iters));
}

int main() {
    uint32_t *out;
    hipMalloc(&out, 256 * 512 * 4);
    for (uint32_t stride : {16u, 144u, 272u}) {
        for (int z = 0; z < (stride == 144 ? 3 : 1); ++z) {
            run<0>("b128", out, stride, z);
            run<1>("read2_b64", out, stride, z);
            run<2>("b64", out, stride, z);
            run<3>("read2_b32", out, stride, z);
        }
    }
    return 0;
}
