// tests/wavesim/gfx950_lds.hpp -- TEST INFRASTRUCTURE ONLY: the wave64 functional model's stand-in for
// ndzip_amd/csrc/gfx950_lds.hpp (the only product header it does not compile as is): the same primitives without the
// VGPR-pinned 32-bit LDS address, the s_waitcnt and the cache policy, which have no meaning on the host.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstring>

namespace ndzip_hip {

#define NDZIP_DEV __device__ __forceinline__

#ifdef WAVESIM_LDSPROF
// the access-profile build (tools/lds_profile.py): 16 bytes travel as ONE 16-byte access (volatile: the host optimiser may not
// split it), so that the load / store hooks see a ds_read_b128 / ds_write_b128 as such
struct alignas(16) vec16 {
    uint32_t w[4];
    vec16() = default;
    vec16(const vec16 &o) { *this = o; }
    vec16 &operator=(const vec16 &o) {
        typedef uint32_t __attribute__((ext_vector_type(4), may_alias)) u32x4;
        *reinterpret_cast<volatile u32x4 *>(w) = *reinterpret_cast<const volatile u32x4 *>(o.w);
        return *this;
    }
};
#else
struct alignas(16) vec16 {
    uint32_t w[4];
};
#endif

NDZIP_DEV vec16 lds_read16(const char *p) {
    if (reinterpret_cast<uintptr_t>(p) % 16 != 0) {  // a misaligned ds_read_b128 is a kernel bug (and 16x slower on gfx950)
        fprintf(stderr, "wavesim: lds_read16 at a misaligned address\n");
        abort();
    }
    return *reinterpret_cast<const vec16 *>(p);
}

// LDS byte address <-> pointer: here the offset inside the running workgroup's LDS array
NDZIP_DEV uint32_t lds_address(const void *p) { return static_cast<uint32_t>(static_cast<const char *>(p) - smem); }
NDZIP_DEV char *lds_pointer(uint32_t address) { return smem + address; }

// v_readfirstlane on hardware: the claim that every active lane holds the same value is CHECKED here (wavesim.cc: uniform_claim
// aborts with the call site when a lane of the wavefront passes another value than the first one did)
template<typename P>
NDZIP_DEV P *scalar_pointer(P *p) {
    return reinterpret_cast<P *>(static_cast<uintptr_t>(wavesim::uniform_claim(static_cast<uint64_t>(reinterpret_cast<uintptr_t>(p)))));
}

NDZIP_DEV uint32_t lane_offset_here(uint32_t bytes) { return bytes; }

NDZIP_DEV int fresh_copy(int x) { return x; }

NDZIP_DEV int wave_uniform(int x) { return static_cast<int>(static_cast<uint32_t>(wavesim::uniform_claim(static_cast<uint32_t>(x)))); }  // (checked, see above)

NDZIP_DEV void lds_reads_issued_before_use(uint32_t (&)[32]) {}  // (instruction scheduling only)

NDZIP_DEV int32_t opaque_vgpr(int32_t x) { return x; }

NDZIP_DEV void registers_complete_here(uint32_t (&)[32]) {}  // (instruction placement only)

NDZIP_DEV void wait_for_own_memory_operations() { std::atomic_thread_fence(std::memory_order_seq_cst); }

NDZIP_DEV uint32_t exchange_performed(uint32_t *p, uint32_t v) { return __atomic_exchange_n(p, v, __ATOMIC_SEQ_CST); }
NDZIP_DEV uint32_t fetch_or_performed(uint32_t *p, uint32_t v) { return __atomic_fetch_or(p, v, __ATOMIC_SEQ_CST); }

NDZIP_DEV vec16 global_load16_once(const void *p) {
    if (reinterpret_cast<uintptr_t>(p) % 16 != 0) {
        fprintf(stderr, "wavesim: global_load16_once at a misaligned address\n");
        abort();
    }
    return *reinterpret_cast<const vec16 *>(p);
}

// The decoder's block load: legal when at least one of the four words lies inside the caller's buffer.  An AddressSanitizer
// build of the model checks that, and hands back junk for the words that are outside -- they must not influence the result.
#if defined(__has_feature)
#if __has_feature(address_sanitizer)
#define WAVESIM_LDS_ASAN 1
extern "C" int __asan_address_is_poisoned(void const volatile *addr);
#endif
#endif
NDZIP_DEV vec16 global_load16_block(const void *p) {
    if (reinterpret_cast<uintptr_t>(p) % 16 != 0) {
        fprintf(stderr, "wavesim: global_load16_block at a misaligned address\n");
        abort();
    }
    vec16 v;
#ifdef WAVESIM_LDS_ASAN
    const char *c = static_cast<const char *>(p);
    int inside = 0;
    for (int i = 0; i < 4; ++i) {
        if (__asan_address_is_poisoned(c + 4 * i) || __asan_address_is_poisoned(c + 4 * i + 3)) {
            v.w[i] = 0xdeadbeefu;
        } else {
            std::memcpy(&v.w[i], c + 4 * i, 4);
            ++inside;
        }
    }
    if (inside == 0) {
        fprintf(stderr, "wavesim: global_load16_block of a block that is entirely outside the buffer\n");
        abort();
    }
#else
    v = *reinterpret_cast<const vec16 *>(p);
#endif
    return v;
}

// the chunk compaction (product: an EXEC-masked store sequence in gfx950 assembly): non-zero words back to back from LDS byte
// address `a` on, returns the address behind the last
NDZIP_DEV uint32_t lds_append_nonzero(uint32_t a, const uint32_t (&w)[32]) {
    for (int i = 0; i < 32; ++i) {
        if (w[i] != 0) {
            *reinterpret_cast<uint32_t *>(lds_pointer(a)) = w[i];
            a += 4;
        }
    }
    return a;
}

// the 64-bit profiles' compaction (product: EXEC-masked assembly): word i kept where bit 31 - i of `flags` is set, stored at the
// XOR-swizzled address of run_layout<uint64_t>, 8 bytes per kept plane
NDZIP_DEV void lds_append_flagged64(uint32_t a, uint32_t flags, const uint32_t (&w)[32]) {
    for (int i = 0; i < 32; ++i) {
        if ((flags >> (31 - i)) & 1u) {
            *reinterpret_cast<uint32_t *>(lds_pointer(a ^ ((a >> 3) & 0x70u))) = w[i];
            a += 8;
        }
    }
}

NDZIP_DEV void lds_append_complete() {}

// inclusive prefix sum of one 64-bit value per lane over the wavefront (product: six v_add_co_u32_dpp + v_addc_co_u32_dpp steps)
NDZIP_DEV void wave_inclusive_scan64(uint32_t &lo, uint32_t &hi) {
    const auto step = [&](int ctrl, int row_mask, bool bound) {
        const uint32_t sl = static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(lo), ctrl, row_mask, 0xf, bound));
        const uint32_t sh = static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(hi), ctrl, row_mask, 0xf, bound));
        const uint64_t r = ((static_cast<uint64_t>(hi) << 32) | lo) + ((static_cast<uint64_t>(sh) << 32) | sl);
        lo = static_cast<uint32_t>(r);
        hi = static_cast<uint32_t>(r >> 32);
    };
    step(0x111, 0xf, true);
    step(0x112, 0xf, true);
    step(0x114, 0xf, true);
    step(0x118, 0xf, true);
    step(0x142, 0xa, false);
    step(0x143, 0xc, false);
}

// lo[j] = odd ? own b[j] : the pair lane's a[j];  hi[j] = odd ? the pair lane's b[j] : own a[j]   (product: v_cndmask_b32_dpp)
NDZIP_DEV void pair_exchange_select4(uint32_t odd_flag, const uint32_t (&a)[4], const uint32_t (&b)[4], uint32_t (&lo)[4], uint32_t (&hi)[4]) {
    for (int j = 0; j < 4; ++j) {
        const uint32_t other_a = static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(a[j]), 0xb1, 0xf, 0xf, true));
        const uint32_t other_b = static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(b[j]), 0xb1, 0xf, 0xf, true));
        lo[j] = odd_flag ? b[j] : other_a;
        hi[j] = odd_flag ? other_b : a[j];
    }
}

// v += row_shr:D(v) for eight 64-bit values as (lo, hi) pairs (product: v_add_co_u32_dpp + v_addc_co_u32_dpp in assembly)
template<int D>
NDZIP_DEV void row_scan_step64(uint32_t (&lo)[8], uint32_t (&hi)[8]) {
    for (int j = 0; j < 8; ++j) {
        const uint32_t sl = static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(lo[j]), 0x110 + D, 0xf, 0xf, true));
        const uint32_t sh = static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(hi[j]), 0x110 + D, 0xf, 0xf, true));
        const uint64_t r = ((static_cast<uint64_t>(hi[j]) << 32) | lo[j]) + ((static_cast<uint64_t>(sh) << 32) | sl);
        lo[j] = static_cast<uint32_t>(r);
        hi[j] = static_cast<uint32_t>(r >> 32);
    }
}

}  // namespace ndzip_hip
