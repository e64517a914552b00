// ndzip_amd/csrc/stages_capi.hip -- the stage entry point of include/ndzip_hip_stages.h: parity-test hooks that run ONE
// hypercube through the device functions the production kernels call (mirror of the reference's stage tests,
// src/test/codec_profile_test.inl:514-549, :552-729, :735-801, :889-947).  Built into libndzip_hip_stages.so with
// stages_f32.hip / stages_f64.hip; the product library (libndzip_hip.so) contains neither this function nor a stage kernel.

#include <hip/hip_runtime.h>

#include <cstdio>
#include <string>

#define NDZIP_HIP_BUILD 1
#include "../../include/ndzip_hip_stages.h"
#include "capi_common.hpp"
#include "codec_launch.hpp"

extern "C" {

const char *ndzip_hip_stages_last_error(void) { return g_last_error.c_str(); }

// where the launch epoch sits in a compressor's scratch (codec_launch.hpp: 16 reserved descriptors, then the ticket lines) -- so that
// a white-box test pokes the word the kernels read, whatever the layout constants are
uint32_t ndzip_hip_debug_scratch_epoch_offset(void) { return 16u * static_cast<uint32_t>(sizeof(tile_desc)) + epoch_word * 4u; }

int ndzip_hip_debug_stage(int stage, int dtype, int dims, const uint32_t *extent, uint32_t hc, const void *d_in, void *d_out,
        uint32_t *d_out_len, uint32_t n, void *hip_stream) {
    if (!valid_dtype(dtype) || !valid_dims(dims)) return fail(NDZIP_HIP_ERR_INVALID_ARGUMENT, "invalid argument");
    if (int s = ensure_device(nullptr)) return s;
    uint32_t one[3] = {side_for_dims(dims), side_for_dims(dims), side_for_dims(dims)};
    const grid_geom gg = make_geom(dims, extent ? extent : one);
    const bool inverse = stage == debug_inverse_transform || stage == debug_inverse_transform_wide;
    const void *array = stage == debug_forward_transform ? d_in : inverse ? d_out : nullptr;
    const bool aligned = array ? is_aligned(dtype, gg, array) : true;
    if ((stage == debug_decode_residuals_wide || stage == debug_inverse_transform_wide) && dtype != NDZIP_HIP_F64) {
        return fail(NDZIP_HIP_ERR_INVALID_ARGUMENT, "stages 8 / 9 are the 256-work-item decoder of 64-bit profiles");
    }
    if ((stage == debug_forward_transform || inverse) && hc >= gg.nhc) {
        return fail(NDZIP_HIP_ERR_INVALID_ARGUMENT, "hypercube index out of range");
    }
    hipStream_t s = static_cast<hipStream_t>(hip_stream);
    HIP_TRY(dtype == NDZIP_HIP_F32 ? launch_debug_stage<float>(stage, dims, gg, hc, d_in, d_out, d_out_len, n, aligned, s)
                                   : launch_debug_stage<double>(stage, dims, gg, hc, d_in, d_out, d_out_len, n, aligned, s));
    return NDZIP_HIP_OK;
}


}  // extern "C"
