// tests/wavesim/hip/hip_runtime.h -- TEST INFRASTRUCTURE ONLY.  Functional model of the gfx950 execution environment
// that the kernels under ndzip_amd/csrc are written for, so that the CPU test suite (no GPU in the authoring container)
// can execute the SAME kernel sources, work-item by work-item, and compare them bit for bit with the oracle.
//
// This is not a back-end and nothing under ndzip_amd/ knows it exists: tests/wavesim/build.py compiles the product's
// .hip/.inl/.hpp files as host C++ against this header (found as <hip/hip_runtime.h>) into
// tests/wavesim/libndzip_hip_wavesim.so, which only tests/test_wavesim_*.py load.  It models WHAT the hardware computes,
// not how fast: a workgroup is an OS thread, each of its work-items a fiber; wave64 cross-lane operations (DPP, shuffles,
// ballots) rendezvous the 64 fibers of a wavefront; LDS is a per-workgroup array; "device memory" is the host heap and
// agent-scope atomics are host atomics.  What it cannot show: timing, LDS bank conflicts, cache (in)coherence between XCDs.
//
// Semantics implemented from the ISA as used by the kernels:
//   v_perm_b32      D.byte[i] = sel.byte[i] in 0..3 ? S1.byte[sel] : 4..7 ? S0.byte[sel-4] : 12 ? 0x00 : >= 13 ? 0xff
//   v_alignbit_b32  ({S0,S1} >> S2[4:0])[31:0]
//   DPP             quad_perm, row_shl/shr/ror, wave_shl/shr/rol/ror:1, row_mirror, row_half_mirror, row_bcast15/31 with
//                   row_mask / bank_mask / bound_ctrl: a lane whose row or bank is masked keeps `old`; a lane whose source
//                   is out of range gets 0 with bound_ctrl and keeps `old` without
#pragma once

#include <atomic>
#include <chrono>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>

// ---- qualifiers ---------------------------------------------------------------------------------------------------
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ thread_local

// ---- the model ----------------------------------------------------------------------------------------------------
namespace wavesim {

struct uint3_ {
    unsigned x, y, z;
};

struct lane_ctx;  // one work-item (fiber)

lane_ctx *self();  // the running work-item
unsigned lane_thread_idx();
unsigned lane_block_idx();
unsigned lane_block_dim();
unsigned lane_grid_dim();

void barrier();                                  // __syncthreads
uint64_t wave_exchange(uint64_t v, int src);     // every lane deposits v, gets lane src's (src < 0 or > 63: own)
uint64_t wave_ballot(bool pred);
uint32_t update_dpp(uint32_t old, uint32_t src, unsigned ctrl, unsigned row_mask, unsigned bank_mask, bool bound_ctrl);
void sleep_hint();                               // s_sleep: lets other workgroups (OS threads) and fibers run
uint64_t uniform_claim(uint64_t v);              // v_readfirstlane of a value claimed wave-uniform: aborts unless every lane passes the same

struct launch_cfg {
    unsigned grid, block;
    size_t lds_bytes;          // dynamic LDS the launch asked for
    char *(*lds_base)();       // the calling translation unit's LDS array of the CURRENT thread (it is thread_local)
};
constexpr size_t lds_capacity = 160 * 1024;
// runs fn(arg) once per work-item of grid x block; workgroups are co-resident up to WAVESIM_MAX_RESIDENT (default 32)
void run_grid(launch_cfg cfg, void (*fn)(void *), void *arg);
// Launch interception (tests/gfx950_exec.py: the built gfx950 code object executed instead of the host-compiled kernel): called
// with the kernel's host function, the launch configuration and the explicit kernel arguments packed as the HSA kernarg ABI lays
// them out (each at its natural alignment); returns true when it has run the grid itself.  Not set = the model runs it.
using launch_hook_t = int (*)(const void *kernel, unsigned grid, unsigned block, unsigned lds_bytes, const void *args, unsigned args_bytes);
launch_hook_t launch_hook();

}  // namespace wavesim

struct wavesim_idx {
    struct X {
        unsigned (*get)();
        operator unsigned() const { return get(); }
    } x;
};
static const wavesim_idx threadIdx{{&wavesim::lane_thread_idx}};
static const wavesim_idx blockIdx{{&wavesim::lane_block_idx}};
static const wavesim_idx blockDim{{&wavesim::lane_block_dim}};
static const wavesim_idx gridDim{{&wavesim::lane_grid_dim}};

// ---- LDS: `extern __shared__ char smem[]` inside a kernel resolves to this per-workgroup-thread array ------------------
namespace ndzip_hip {
namespace {
alignas(128) thread_local char smem[wavesim::lds_capacity];
}
}  // namespace ndzip_hip

// ---- device intrinsics --------------------------------------------------------------------------------------------
inline void __syncthreads() { wavesim::barrier(); }
inline void __threadfence() { std::atomic_thread_fence(std::memory_order_seq_cst); }

template<typename T>
inline T wavesim_shfl_src(T v, int src) {
    static_assert(std::is_trivially_copyable<T>::value && sizeof(T) <= 8, "shuffle of up to 8 bytes");
    uint64_t bits = 0;
    std::memcpy(&bits, &v, sizeof(T));
    bits = wavesim::wave_exchange(bits, src);
    T out;
    std::memcpy(&out, &bits, sizeof(T));
    return out;
}
inline int wavesim_lane() { return static_cast<int>(wavesim::lane_thread_idx() & 63u); }
// (the index arithmetic of HIP's own definitions, amd_warp_functions.h: a value never crosses a `width`-lane segment)
template<typename T>
inline T __shfl(T v, int src, int width = 64) {
    const int self = wavesim_lane();
    return wavesim_shfl_src(v, (src & (width - 1)) + (self & ~(width - 1)));
}
template<typename T>
inline T __shfl_up(T v, unsigned d, int width = 64) {
    const int self = wavesim_lane();
    const int index = self - static_cast<int>(d);
    return wavesim_shfl_src(v, index < (self & ~(width - 1)) ? self : index);
}
template<typename T>
inline T __shfl_xor(T v, int mask, int width = 64) {
    const int self = wavesim_lane();
    const int index = self ^ mask;
    return wavesim_shfl_src(v, index >= ((self + width) & ~(width - 1)) ? self : index);
}
inline unsigned long long __ballot(bool pred) { return wavesim::wave_ballot(pred); }

inline uint32_t wavesim_perm(uint32_t s0, uint32_t s1, uint32_t sel) {
    uint32_t out = 0;
    for (int i = 0; i < 4; ++i) {
        const uint32_t c = (sel >> (8 * i)) & 0xffu;
        uint32_t b;
        if (c <= 3) {
            b = (s1 >> (8 * c)) & 0xffu;
        } else if (c <= 7) {
            b = (s0 >> (8 * (c - 4))) & 0xffu;
        } else if (c <= 11) {  // 8: S1[15], 9: S1[31], 10: S0[15], 11: S0[31], replicated
            const uint32_t word = c >= 10 ? s0 : s1;
            const int bit = (c & 1) ? 31 : 15;
            b = ((word >> bit) & 1u) ? 0xffu : 0u;
        } else if (c == 12) {
            b = 0;
        } else {
            b = 0xffu;
        }
        out |= b << (8 * i);
    }
    return out;
}
#define __builtin_amdgcn_perm(s0, s1, sel) wavesim_perm((s0), (s1), (sel))
#define __builtin_amdgcn_alignbit(hi, lo, sh) \
    static_cast<uint32_t>(((static_cast<uint64_t>(hi) << 32) | static_cast<uint64_t>(lo)) >> ((sh) & 31u))
#define __builtin_amdgcn_update_dpp(old, src, ctrl, rmask, bmask, bctrl) \
    static_cast<int>(wavesim::update_dpp(static_cast<uint32_t>(old), static_cast<uint32_t>(src), (ctrl), (rmask), (bmask), (bctrl)))
#define __builtin_amdgcn_readlane(v, lane) static_cast<int>(wavesim_shfl_src(static_cast<uint32_t>(v), (lane) & 63))
#define __builtin_amdgcn_sched_barrier(x) ((void) 0)
#define __builtin_amdgcn_s_sleep(x) wavesim::sleep_hint()

inline uint32_t __umulhi(uint32_t a, uint32_t b) { return static_cast<uint32_t>((static_cast<uint64_t>(a) * b) >> 32); }

#define __HIP_MEMORY_SCOPE_AGENT 4
#define __hip_atomic_load(p, order, scope) __atomic_load_n((p), (order))
#define __hip_atomic_store(p, v, order, scope) __atomic_store_n((p), (v), (order))
#define __hip_atomic_exchange(p, v, order, scope) __atomic_exchange_n((p), (v), (order))
#define __hip_atomic_fetch_or(p, v, order, scope) __atomic_fetch_or((p), (v), (order))
template<typename T>
inline T atomicAdd(T *p, T v) {
    return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST);
}
template<typename T>
inline T atomicOr(T *p, T v) {
    return __atomic_fetch_or(p, v, __ATOMIC_SEQ_CST);
}

#ifdef WAVESIM_LDSPROF
// the access-profile build: an 8-byte value travels as ONE 8-byte access (volatile: the host optimiser may not split it into the
// two dwords the fields are), so that the hooks see the width the GPU instruction has
struct alignas(8) uint2 {
    uint32_t x, y;
    uint2() = default;
    uint2(uint32_t a, uint32_t b) : x(a), y(b) {}
    uint2(const uint2 &o) { *this = o; }
    uint2 &operator=(const uint2 &o) {
        typedef uint64_t __attribute__((may_alias)) u64;
        *reinterpret_cast<volatile u64 *>(this) = *reinterpret_cast<const volatile u64 *>(&o);
        return *this;
    }
};
#else
struct uint2 {
    uint32_t x, y;
};
#endif
inline uint2 make_uint2(uint32_t x, uint32_t y) { return uint2{x, y}; }

// ---- host runtime: "device memory" is the heap, streams are synchronous -------------------------------------------------
typedef int hipError_t;
constexpr hipError_t hipSuccess = 0;
constexpr hipError_t hipErrorInvalidValue = 1;
constexpr hipError_t hipErrorOutOfMemory = 2;
inline const char *hipGetErrorString(hipError_t e) { return e == hipSuccess ? "success" : e == hipErrorOutOfMemory ? "out of memory" : "invalid value"; }
inline hipError_t hipGetLastError() { return hipSuccess; }

typedef struct wavesim_stream *hipStream_t;
struct wavesim_event {
    std::chrono::steady_clock::time_point t;
};
typedef wavesim_event *hipEvent_t;
struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
enum hipMemcpyKind { hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
enum hipDeviceAttribute_t { hipDeviceAttributeMultiprocessorCount, hipDeviceAttributeNumberOfXccs };
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize };
constexpr unsigned hipStreamNonBlocking = 1;
constexpr unsigned hipHostMallocDefault = 0;
struct hipDeviceProp_t {
    char gcnArchName[256];
};

inline int wavesim_env_int(const char *name, int dflt) {
    const char *e = getenv(name);
    return e && *e ? atoi(e) : dflt;
}
inline hipError_t hipGetDeviceCount(int *n) {
    *n = 1;
    return hipSuccess;
}
inline hipError_t hipSetDevice(int d) { return d == 0 ? hipSuccess : hipErrorInvalidValue; }  // (one "device")
inline hipError_t hipGetDevice(int *d) {
    *d = 0;
    return hipSuccess;
}
inline hipError_t hipDeviceGetAttribute(int *v, hipDeviceAttribute_t what, int) {
    if (what == hipDeviceAttributeNumberOfXccs) {
        *v = wavesim_env_int("WAVESIM_XCDS", 8);  // (only an order of the decoder's tiles: any value must give the same bytes)
    } else {
        *v = wavesim_env_int("WAVESIM_CUS", 2);  // a persistent grid of 2 x (workgroups per CU) workgroups
    }
    return hipSuccess;
}
inline hipError_t hipGetDeviceProperties(hipDeviceProp_t *p, int) {
    snprintf(p->gcnArchName, sizeof p->gcnArchName, "gfx950-wavesim");
    return hipSuccess;
}
// Device memory: exactly the bytes asked for (an AddressSanitizer build of the model then traps the first byte beyond), and
// filled with junk -- hipMalloc does not hand out zeroed memory and the kernels must not count on it.
inline hipError_t hipMalloc(void **p, size_t bytes) {
    *p = nullptr;
    if (posix_memalign(p, 256, bytes ? bytes : 1) != 0) return hipErrorOutOfMemory;
    memset(*p, 0xCD, bytes);
    return hipSuccess;
}
inline hipError_t hipFree(void *p) {
    free(p);
    return hipSuccess;
}
inline hipError_t hipHostMalloc(void **p, size_t bytes, unsigned) { return hipMalloc(p, bytes); }
inline hipError_t hipHostFree(void *p) { return hipFree(p); }
inline hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) {
    memcpy(d, s, n);
    return hipSuccess;
}
inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t) {
    memcpy(d, s, n);
    return hipSuccess;
}
inline hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t) {
    memset(d, v, n);
    return hipSuccess;
}
typedef void *hipDeviceptr_t;
inline hipError_t hipMemsetD32Async(hipDeviceptr_t d, int v, size_t count, hipStream_t) {
    for (size_t i = 0; i < count; ++i) static_cast<int *>(d)[i] = v;
    return hipSuccess;
}
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
inline hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) {
    *s = nullptr;
    return hipSuccess;
}
inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
inline hipError_t hipEventCreate(hipEvent_t *e) {
    *e = new wavesim_event{};
    return hipSuccess;
}
inline hipError_t hipEventDestroy(hipEvent_t e) {
    delete e;
    return hipSuccess;
}
inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t) {
    e->t = std::chrono::steady_clock::now();
    return hipSuccess;
}
inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b) {
    *ms = std::chrono::duration<float, std::milli>(b->t - a->t).count();
    return hipSuccess;
}
inline hipError_t hipFuncSetAttribute(const void *, hipFuncAttribute, int) { return hipSuccess; }
template<typename K>
inline hipError_t hipOccupancyMaxActiveBlocksPerMultiprocessor(int *n, K, int, size_t) {
    *n = wavesim_env_int("WAVESIM_BLOCKS_PER_CU", 2);
    return hipSuccess;
}

// ---- kernel launch: bind the arguments, run every work-item of the grid ------------------------------------------------
#include <tuple>
#include <utility>

template<typename T>
inline void wavesim_pack_kernarg(unsigned char *buf, size_t &at, const T &v) {
    at = (at + alignof(T) - 1) / alignof(T) * alignof(T);
    std::memcpy(buf + at, &v, sizeof(T));
    at += sizeof(T);
}

template<typename... Params, typename... Args>
inline void wavesim_launch(void (*kernel)(Params...), dim3 grid, dim3 block, size_t lds_bytes, Args &&...args) {
    std::tuple<std::decay_t<Params>...> bound{static_cast<std::decay_t<Params>>(std::forward<Args>(args))...};
    if (const auto hook = wavesim::launch_hook()) {
        alignas(16) unsigned char buf[(sizeof(std::decay_t<Params>) + ... + 0) + 16 * sizeof...(Params) + 16] = {};
        size_t at = 0;
        std::apply([&](const auto &...v) { (wavesim_pack_kernarg(buf, at, v), ...); }, bound);
        if (hook(reinterpret_cast<const void *>(kernel), grid.x, block.x, static_cast<unsigned>(lds_bytes), buf, static_cast<unsigned>(at))) return;
    }
    struct thunk_t {
        void (*kernel)(Params...);
        std::tuple<std::decay_t<Params>...> *bound;
    } thunk{kernel, &bound};
    wavesim::run_grid(
            wavesim::launch_cfg{grid.x, block.x, lds_bytes, +[]() -> char * { return ndzip_hip::smem; }},
            [](void *p) {
                auto *t = static_cast<thunk_t *>(p);
                std::apply(t->kernel, *t->bound);
            },
            &thunk);
}
#define hipLaunchKernelGGL(kernel, grid, block, smem_bytes, stream, ...) \
    ((void) (stream), wavesim_launch(kernel, grid, block, static_cast<size_t>(smem_bytes), __VA_ARGS__))
