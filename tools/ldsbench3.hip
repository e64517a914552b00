// tools/ldsbench3.hip -- the Lorenzo stencil's exact LDS read patterns (padded chunk layout), ds_read_b128, scanning the
// 16-byte slot (mod 256) of the shared zero block that out-of-cube lanes read.  1 wavefront per launch, 16 reads.
// argv: none.  Launch order: for cfg in {f32-3d, f64-3d, f32-2d, f64-2d}: for pat: for wave: for slot 0..15.
#include <hip/hip_runtime.h>
#include <cstdio>

__device__ inline uint32_t pattern_addr(int cfg, int pat, int t, uint32_t zero) {
    const uint32_t es = (cfg & 1) ? 8 : 4, stride = 32 * es + 16, half = 16 * es;
    if (cfg < 2) {  // 3D: chunk = rows (z, y0), (z, y0+1)
        const int z = t >> 3, yp = t & 7;
        switch (pat) {
            case 0: return t * stride;
            case 1: return yp > 0 ? (t - 1) * stride + half : zero;
            case 2: return z > 0 ? (t - 8) * stride : zero;
            default: return (z > 0 && yp > 0) ? (t - 9) * stride + half : zero;
        }
    } else {  // 2D: chunk = half a row; up = chunk t-2
        const int y = t >> 1;
        switch (pat) {
            case 0: return t * stride;
            default: return y > 0 ? (t - 2) * stride : zero;
        }
    }
}

__global__ void __launch_bounds__(64) k(uint32_t *out, int cfg, int pat, int wave, uint32_t slot) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const uint32_t lane = threadIdx.x;
    for (uint32_t i = lane; i < 16384; i += 64) reinterpret_cast<uint32_t *>(smem)[i] = i;
    __syncthreads();
    const uint32_t base = static_cast<uint32_t>(reinterpret_cast<uintptr_t>((const __attribute__((address_space(3))) char *) smem));
    const uint32_t addr = base + pattern_addr(cfg, pat, wave * 64 + lane, 128 * 272 + slot * 16);
    uint32_t acc = 0;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        uint32_t a = addr + (j & 3) * 16;
        uint4 v;
        asm volatile("ds_read_b128 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
        acc += v.x + v.w;
    }
    out[lane] = acc;
}

int main() {
    uint32_t *out;
    hipMalloc(&out, 4096);
    for (int cfg = 0; cfg < 4; ++cfg)
        for (int pat = 1; pat < (cfg < 2 ? 4 : 2); ++pat)
            for (int wave = 0; wave < 2; ++wave)
                for (uint32_t slot = 0; slot < 16; ++slot) {
                    hipLaunchKernelGGL(k, dim3(1), dim3(64), 65536, 0, out, cfg, pat, wave, slot);
                    hipDeviceSynchronize();
                }
    return 0;
}
