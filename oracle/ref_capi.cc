// oracle/ref_capi.cc -- extern "C" doorway into the REAL reference (celerity/ndzip serial CPU path).
//
// TEST INFRASTRUCTURE ONLY.  This file is ours; everything it calls is the reference's own code,
// compiled by oracle/Makefile from the sources where they lie under /root/reference (nothing is copied
// into this repository, and the resulting oracle/_ref/libndzip_ref.so is git-ignored).  It is used to
// (1) validate the plain-C restatement in ndzip_oracle.c, (2) generate the golden fixtures under
// tests/golden/, and (3) optionally serve as the `cpu_baseline` of kind "reference" in bench.py.
//
// Only the serial path (make_cpu_offloader<T>(dims, 1)) is reachable: the reference's OpenMP path needs
// Boost, which this image lacks, so it is treated as unbuildable.

#include <cassert>
#include <cstdint>
#include <stdexcept>

#include <ndzip/ndzip.hh>
#include <ndzip/offload.hh>

namespace {

ndzip::extent make_extent(int dims, const uint32_t *e) {
    ndzip::extent ext(dims);
    for (int d = 0; d < dims; ++d) ext[d] = e[d];
    return ext;
}

template<typename T>
int64_t ref_compress(int dims, const uint32_t *e, const T *in, ndzip::compressed_type<T> *out) {
    try {
        auto off = ndzip::make_cpu_offloader<T>(dims, 1);
        return off->compress(in, make_extent(dims, e), out);
    } catch (...) { return -1; }
}

template<typename T>
int64_t ref_decompress(int dims, const uint32_t *e, const ndzip::compressed_type<T> *in, uint32_t len, T *out) {
    try {
        auto off = ndzip::make_cpu_offloader<T>(dims, 1);
        return off->decompress(in, len, out, make_extent(dims, e));
    } catch (...) { return -1; }
}

}  // namespace

extern "C" {

__attribute__((visibility("default"))) int64_t
ndzip_ref_compress_f32(int dims, const uint32_t *e, const float *in, uint32_t *out) {
    return ref_compress<float>(dims, e, in, out);
}

__attribute__((visibility("default"))) int64_t
ndzip_ref_compress_f64(int dims, const uint32_t *e, const double *in, uint64_t *out) {
    return ref_compress<double>(dims, e, in, out);
}

__attribute__((visibility("default"))) int64_t
ndzip_ref_decompress_f32(int dims, const uint32_t *e, const uint32_t *in, uint32_t len, float *out) {
    return ref_decompress<float>(dims, e, in, len, out);
}

__attribute__((visibility("default"))) int64_t
ndzip_ref_decompress_f64(int dims, const uint32_t *e, const uint64_t *in, uint32_t len, double *out) {
    return ref_decompress<double>(dims, e, in, len, out);
}

__attribute__((visibility("default"))) uint64_t ndzip_ref_compressed_length_bound(int bits, int dims, const uint32_t *e) {
    return bits == 32 ? ndzip::compressed_length_bound<float>(make_extent(dims, e))
                      : ndzip::compressed_length_bound<double>(make_extent(dims, e));
}

}  // extern "C"
