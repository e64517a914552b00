// float instantiation of the STAGE kernels only (parity-test hooks, libndzip_hip_stages.so): the same device functions the
// production kernels of kernels_f32.hip call, one hypercube at a time (see codec_launch.inl, NDZIP_STAGE_KERNELS)
#define NDZIP_T float
#define NDZIP_STAGE_KERNELS 1
#include "codec_launch.inl"
