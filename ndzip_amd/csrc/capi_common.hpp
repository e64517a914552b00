// ndzip_amd/csrc/capi_common.hpp -- host helpers shared by the two C-ABI translation units: capi.hip (the product library,
// include/ndzip_hip.h) and stages_capi.hip (the parity-test hooks, include/ndzip_hip_stages.h).  Everything sits in an anonymous
// namespace on purpose: each library keeps its own last-error string.
#pragma once

#include <hip/hip_runtime.h>

#include <string>

#include "../../include/ndzip_hip.h"
#include "codec_common.hpp"

using namespace ndzip_hip;

namespace {

thread_local std::string g_last_error;

int fail(int status, const std::string &msg) {
    g_last_error = msg;
    return status;
}

int fail_hip(hipError_t e, const char *what) {
    return fail(NDZIP_HIP_ERR_RUNTIME, std::string(what) + ": " + hipGetErrorString(e));
}

#define HIP_TRY(expr)                                          \
    do {                                                       \
        hipError_t e_ = (expr);                                \
        if (e_ != hipSuccess) return fail_hip(e_, #expr);      \
    } while (0)

bool valid_dtype(int dtype) { return dtype == NDZIP_HIP_F32 || dtype == NDZIP_HIP_F64; }
bool valid_dims(int dims) { return dims >= 1 && dims <= 3; }
size_t word_bytes(int dtype) { return dtype == NDZIP_HIP_F32 ? 4 : 8; }
uint32_t header_words_for(int dtype, uint32_t nhc) { return dtype == NDZIP_HIP_F32 ? nhc : (nhc + 1) / 2; }

// The calling thread's current device: its ordinal, compute units and accelerator complexes (XCDs, each with its own L2;
// 8 on an MI355X in SPX mode -- asked of the runtime, not assumed: a partitioned device reports fewer).
int ensure_device(int *num_cus, int *device = nullptr, int *num_xcds = nullptr) {
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) {
        (void) hipGetLastError();
        return fail(NDZIP_HIP_ERR_NO_DEVICE, "no HIP device visible: the ndzip HIP back-end has no CPU fallback");
    }
    int dev = 0;
    if (num_cus || device || num_xcds) HIP_TRY(hipGetDevice(&dev));
    if (device) *device = dev;
    if (num_cus) HIP_TRY(hipDeviceGetAttribute(num_cus, hipDeviceAttributeMultiprocessorCount, dev));
    if (num_xcds) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeNumberOfXccs, dev) != hipSuccess || n < 1) {
            (void) hipGetLastError();
            n = 1;  // unknown: tiles in plain order (only locality depends on it)
        }
        *num_xcds = n;
    }
    return NDZIP_HIP_OK;
}

// 16-byte vector path is legal when the base pointer and every hypercube-row start are 16-byte aligned
bool is_aligned(int dtype, const grid_geom &gg, const void *data) {
    const uint64_t ve = 16 / word_bytes(dtype);
    if (reinterpret_cast<uintptr_t>(data) % 16 != 0) return false;
    for (uint32_t d = 0; d + 1 < gg.dims; ++d) {
        if (gg.stride[d] % ve != 0) return false;
    }
    return true;
}

}  // namespace
