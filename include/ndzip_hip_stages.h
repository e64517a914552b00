/* include/ndzip_hip_stages.h -- parity-test hooks of the MI355X back-end, exported by libndzip_hip_stages.so (NOT by the product
 * library libndzip_hip.so, which contains no stage kernel): ONE hypercube at a time through the very device functions the production
 * kernels call.  No counterpart in the reference's public interface; what it mirrors is the reference's own stage-level test
 * strategy (src/test/codec_profile_test.inl:514-549 forward transform, :552-729 residual encoding, :735-801 chunk decoding,
 * :889-947 inverse transform; src/test/cuda_bits_test.cu:94-114 the device-wide scan; src/test/codec_generic_test.cc:65-81 the bit
 * transpose).  Status codes and handles: include/ndzip_hip.h. */
#ifndef NDZIP_HIP_STAGES_H
#define NDZIP_HIP_STAGES_H

#include "ndzip_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* message of the last failed call of THIS library on the calling thread (the product library keeps its own) */
NDZIP_HIP_API const char *ndzip_hip_stages_last_error(void);

/* stage: 0 forward transform of hypercube `hc` of the device array `d_in` -> 4096 residual words in `d_out`
 *        1 encode 4096 residual words -> encoded run in `d_out` (4096 + 4096/B words capacity), *d_out_len words
 *        2 decode an encoded run -> 4096 residual words
 *        3 inverse transform of 4096 residual words -> hypercube `hc` of the device array `d_out`
 *        4 / 5 32x32 bit transposes of `n` blocks of 32 uint32 (v_perm network / shift-mask network)
 *        8 / 9 stages 2 / 3 through the 256-work-item decoder of the 64-bit profiles (dtype NDZIP_HIP_F64 only)
 *        6 the wave64 scan and sum (DPP): `n` uint32 (a multiple of 64) -> per wavefront the inclusive prefix sums, then
 *          n / 64 wave totals behind them (`d_out` holds n + n / 64 words)
 *        7 the device-wide scan on its own: `n` uint32 tile lengths -> their n exclusive prefix sums, the total, the error word
 *          (`d_out` holds n + 2 words); `hc` = workgroups of the persistent grid (0 = what a production launch would use) */
NDZIP_HIP_API int ndzip_hip_debug_stage(int stage, int dtype, int dims, const uint32_t *extent, uint32_t hc, const void *d_in,
        void *d_out, uint32_t *d_out_len, uint32_t n, void *hip_stream);

/* byte offset of the launch-epoch word inside a compressor's scratch (ndzip_amd/csrc/codec_launch.hpp: epoch_word); for the
 * white-box test of the epoch's wrap-around */
NDZIP_HIP_API uint32_t ndzip_hip_debug_scratch_epoch_offset(void);

#ifdef __cplusplus
}
#endif

#endif /* NDZIP_HIP_STAGES_H */
