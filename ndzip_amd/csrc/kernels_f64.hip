// double instantiation of the codec kernels (see codec_launch.inl)
#define NDZIP_T double
#include "codec_launch.inl"
