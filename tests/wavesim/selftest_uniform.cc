// tests/wavesim/selftest_uniform.cc -- TEST INFRASTRUCTURE ONLY: the model's check of wave_uniform / scalar_pointer claims, on
// kernels that make a true claim, a false one, and a true one under divergent control flow (tests/test_wavesim_model.py).
#include <hip/hip_runtime.h>

#include "gfx950_lds.hpp"

using namespace ndzip_hip;

__global__ void claims(int mode, uint32_t *out) {
    const int tid = static_cast<int>(threadIdx.x);
    int v = 0;
    if (mode == 0) {
        v = wave_uniform(tid >> 6);  // true: the wave index
    } else if (mode == 1) {
        v = wave_uniform(tid >> 5);  // false: differs between the two halves of a wavefront
    } else if (mode == 2) {
        // true, but only some lanes execute it, and a different number of times each (v_readfirstlane under a partial EXEC mask)
        for (int i = 0; i < (tid & 3); ++i) v += wave_uniform(100 + i + (tid >> 6));
    } else if (mode == 3) {
        uint32_t *p = scalar_pointer(out + (tid >> 6) * 64);  // true
        v = static_cast<int>(p - out);
    } else if (mode == 4) {
        uint32_t *p = scalar_pointer(out + tid);  // false: a per-lane pointer
        v = static_cast<int>(p - out);
    } else {
        // the same call site before and after a wave operation with different (each time uniform) values: the execution count
        // restarts where the wavefront met
        v = 0;
        for (int i = 0; i < 3; ++i) {
            v += wave_uniform(i * 7 + (tid >> 6));
            v += static_cast<int>(__ballot(true) & 1u);
        }
    }
    out[tid] = static_cast<uint32_t>(v);
}

int main(int argc, char **argv) {
    const int mode = argc > 1 ? atoi(argv[1]) : 0;
    uint32_t *out = nullptr;
    if (hipMalloc(reinterpret_cast<void **>(&out), 128 * sizeof(uint32_t)) != hipSuccess) return 2;
    hipLaunchKernelGGL(claims, dim3(2), dim3(128), 0, nullptr, mode, out);
    printf("ok %u %u\n", out[0], out[127]);
    return 0;
}
