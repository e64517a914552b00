// tests/wavesim/gfx950_lds.hpp -- TEST INFRASTRUCTURE ONLY: the wave64 functional model's stand-in for
// ndzip_amd/csrc/gfx950_lds.hpp (the only product header it does not compile as is): the same primitives without the
// VGPR-pinned 32-bit LDS address, the s_waitcnt and the cache policy, which have no meaning on the host.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstring>

namespace ndzip_hip {

#define NDZIP_DEV __device__ __forceinline__

struct alignas(16) vec16 {
    uint32_t w[4];
};

NDZIP_DEV vec16 lds_read16(const char *p) {
    if (reinterpret_cast<uintptr_t>(p) % 16 != 0) {  // a misaligned ds_read_b128 is a kernel bug (and 16x slower on gfx950)
        fprintf(stderr, "wavesim: lds_read16 at a misaligned address\n");
        abort();
    }
    vec16 v;
    std::memcpy(&v, p, sizeof v);
    return v;
}

NDZIP_DEV void lds_reads_issued_before_use(uint32_t (&)[32]) {}  // (instruction scheduling only)

NDZIP_DEV int32_t opaque_vgpr(int32_t x) { return x; }

NDZIP_DEV void wait_for_own_memory_operations() { std::atomic_thread_fence(std::memory_order_seq_cst); }

NDZIP_DEV vec16 global_load16_once(const void *p) {
    if (reinterpret_cast<uintptr_t>(p) % 16 != 0) {
        fprintf(stderr, "wavesim: global_load16_once at a misaligned address\n");
        abort();
    }
    vec16 v;
    std::memcpy(&v, p, sizeof v);
    return v;
}

}  // namespace ndzip_hip
