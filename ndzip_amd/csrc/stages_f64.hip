// double instantiation of the STAGE kernels only (parity-test hooks, libndzip_hip_stages.so; see stages_f32.hip)
#define NDZIP_T double
#define NDZIP_STAGE_KERNELS 1
#include "codec_launch.inl"
