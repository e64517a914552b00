// tools/ldsbench.hip -- isolate LDS bank conflicts of single access patterns (run under rocprofv3 --pmc SQ_LDS_BANK_CONFLICT ...)
#include <hip/hip_runtime.h>
#include <cstdio>
#include "codec_kernels.hpp"
using namespace ndzip_hip;
using L = lds_layout<uint32_t>;

template<int Mode>
__global__ void __launch_bounds__(128) k(uint32_t *out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int t = threadIdx.x;
    vec16 v{{1u * t, 2u * t, 3u * t, 4u * t}};
    uint32_t acc = 0;
    if (Mode == 0) {  // staging writes
        for (int i = 0; i < 8; ++i) lds_write16(smem + L::off((i * 128 + t) * 4), v);
    } else if (Mode == 1) {  // own chunk: 8 x b128 at stride 144 B per lane
        for (int j = 0; j < 8; ++j) { vec16 r = lds_read16(smem + L::off(32 * t) + 16 * j); acc += r.w[0] + r.w[3]; }
    } else if (Mode == 2) {  // same, but contiguous 16 B per lane (no stride)
        for (int j = 0; j < 8; ++j) { vec16 r = lds_read16(smem + 16 * t + 2048 * j); acc += r.w[0] + r.w[3]; }
    } else if (Mode == 3) {  // stride 144 B via b64 x 2
        for (int j = 0; j < 16; ++j) { uint2 r = *reinterpret_cast<const uint2 *>(smem + L::off(32 * t) + 8 * j); acc += r.x + r.y; }
    } else if (Mode == 4) {  // stride 132 B (pad 4 B per 32 values), b128 would be misaligned -> b32 x 4
        for (int j = 0; j < 32; ++j) { acc += *reinterpret_cast<const uint32_t *>(smem + 132 * t + 4 * j); }
    } else if (Mode == 5) {  // stride 80 B per lane, b128
        for (int j = 0; j < 4; ++j) { vec16 r = lds_read16(smem + 80 * t + 16 * j); acc += r.w[0] + r.w[3]; }
    } else if (Mode == 6) {  // stride 272 B per lane, b128 (f64 layout)
        for (int j = 0; j < 4; ++j) { vec16 r = lds_read16(smem + 272 * (t & 63) + 16 * j); acc += r.w[0] + r.w[3]; }
    }
    out[t] = acc;
}

template<int Mode> void run(uint32_t *out) { hipLaunchKernelGGL(k<Mode>, dim3(1), dim3(128), 40000, 0, out); hipDeviceSynchronize(); }
int main() {
    uint32_t *out; hipMalloc(&out, 4096);
    run<0>(out); run<1>(out); run<2>(out); run<3>(out); run<4>(out); run<5>(out); run<6>(out);
    printf("done\n");
    return 0;
}
