// float instantiation of the codec kernels (see codec_launch.inl)
#define NDZIP_T float
#include "codec_launch.inl"
