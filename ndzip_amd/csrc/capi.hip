// ndzip_amd/csrc/capi.hip -- the C ABI declared in include/ndzip_hip.h, plus the border / bookkeeping kernels.
//
// Host logic mirrors the behaviour (not the code) of the reference drivers:
//   cuda_compressor_impl::compress      src/ndzip/cuda_codec.inl:554-603
//   cuda_decompressor_impl::decompress  src/ndzip/cuda_codec.inl:628-652
//   cuda_offloader::do_compress / do_decompress  src/ndzip/cuda_codec.inl:669-761
// There is NO CPU fallback: without a HIP device every compute entry point fails with NDZIP_HIP_ERR_NO_DEVICE.

#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#define NDZIP_HIP_BUILD 1
#include "capi_common.hpp"
#include "codec_launch.hpp"

namespace {

// NDZIP_VERBOSE (any non-empty value), the reference's only tracing switch (src/ndzip/common.hh:630-633): the device-pointer
// entry points report the hypercube count when they enqueue (cuda_codec.inl:565-567; nothing is timed there -- they never
// synchronise), the host-pointer entry points the device pipeline time by events (cuda_codec.inl:688-703, :733-755).
// Goes to stderr, so a stream written to stdout stays intact.  The ONLY environment variable this library reads.
bool verbose() {
    static const bool on = [] {
        const char *e = getenv("NDZIP_VERBOSE");
        return e && *e;
    }();
    return on;
}

// compressed_length_bound (common.cc:31-55), in 64 bits
uint64_t length_bound(int dtype, const grid_geom &gg) {
    const uint64_t B = dtype == NDZIP_HIP_F32 ? 32 : 64;
    return header_words_for(dtype, gg.nhc) + static_cast<uint64_t>(gg.nhc) * (hc_size / B * (B + 1)) + border_count(gg);
}

int check_limits(int dtype, const grid_geom &gg) {
    if (num_elements(gg) > 0xffffffffull) return fail(NDZIP_HIP_ERR_LIMIT, "extent has more than 2^32-1 elements (index_type is uint32_t)");
    if (length_bound(dtype, gg) > 0xffffffffull) return fail(NDZIP_HIP_ERR_LIMIT, "compressed length bound exceeds 2^32-1 words");
    return NDZIP_HIP_OK;
}

// ---- border gather/scatter (reference: compact_border / expand_border, cuda_codec.inl:463-474, :495-504) ------

template<typename W, bool Pack>
__global__ void border_kernel(W *data, W *body, const uint32_t *header, uint32_t nhc, const uint32_t *header_base, border_geom bg,
        uint32_t *out_len, uint32_t len_extra, uint32_t *err, uint32_t body_words) {
    const uint64_t start = nhc ? header[nhc - 1] - (header_base ? *header_base : 0u) : 0u;  // stream<Profile>::border(), common.hh:365
    bool corrupt = false;
    if (!Pack) {
        // unpacking trusts the last header entry only as far as the format allows (same rule as decompress_kernel); a border
        // that cannot be located is written as zeros, with the error word set, rather than left as whatever the buffer held
        constexpr uint64_t B = sizeof(W) * 8;
        const uint64_t lo = static_cast<uint64_t>(nhc) * (hc_size / B), hi = static_cast<uint64_t>(nhc) * (hc_size / B * (B + 1));
        corrupt = start < lo || start > hi || start + bg.count > body_words;
        if (corrupt && blockIdx.x == 0 && threadIdx.x == 0) atomicOr(err, 2u);
    }
    W *border = body + start;
    const uint64_t zpart = bg.cz * bg.per_z;
    const uint64_t tails = bg.cy * bg.tail;
    for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < bg.count;
            i += static_cast<uint64_t>(gridDim.x) * blockDim.x) {
        uint64_t src;
        if (i >= zpart) {
            src = bg.cz * bg.ny * bg.nx + (i - zpart);
        } else {
            const uint64_t z = i / bg.per_z, r = i % bg.per_z;
            if (r < tails) {
                src = (z * bg.ny + r / bg.tail) * bg.nx + bg.cx + r % bg.tail;
            } else {
                src = (z * bg.ny + bg.cy) * bg.nx + (r - tails);
            }
        }
        if (Pack) {
            border[i] = data[src];
        } else {
            data[src] = corrupt ? W{0} : border[i];
        }
    }
    // with zero hypercubes nobody else writes the stream length (store_stream_length, cuda_codec.inl:507-511)
    if (Pack && out_len && nhc == 0 && blockIdx.x == 0 && threadIdx.x == 0) *out_len = len_extra;
}

__global__ void store_length_kernel(uint32_t *out_len, uint32_t value) { *out_len = value; }

__global__ void offset_header_kernel(uint32_t *header, uint32_t count, uint32_t base) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < count; i += gridDim.x * blockDim.x) header[i] += base;
}

__global__ void offset_header_device_kernel(uint32_t *header, uint32_t count, const uint32_t *base) {
    const uint32_t b = *base;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < count; i += gridDim.x * blockDim.x) header[i] += b;
}

// base of shard `rank` = sum over lower ranks of (body words incl. border - border words), from the all-gathered lengths;
// added to the shard's header entries and left in *base_out for the decoder (one launch instead of a dozen tensor ops).
// Offsets of the stream format are index_type = uint32 (include/ndzip/ndzip.hh:20, common.hh:342-347): `world` legal shards can
// add up to hypercube runs the format cannot address.  The sums are taken in 64 bits over ALL shards, so every rank of the plan
// sets err_offset_overflow in its error word for the same inputs (SURVEY 8e "keep u64 internally and check overflow").
constexpr uint32_t err_offset_overflow_bit = 4u;
__global__ void offset_header_gathered_kernel(uint32_t *header, uint32_t count, const uint32_t *lengths, const uint32_t *borders,
        uint32_t rank, uint32_t world, uint32_t *base_out, uint32_t *err) {
    uint64_t base = 0, total = 0;
    for (uint32_t r = 0; r < world; ++r) {
        if (r == rank) base = total;
        total += static_cast<uint64_t>(lengths[r]) - borders[r];
    }
    // A plan whose runs do not fit the format's 32-bit offsets has no global header: the entries stay LOCAL and the base 0 -- a
    // state the rank can still decode its own slab from -- instead of entries wrapped modulo 2^32, and the error word says why.
    const bool overflow = total > 0xffffffffull;
    const uint32_t base32 = overflow ? 0u : static_cast<uint32_t>(base);
    if (base32 != 0) {
        for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < count; i += gridDim.x * blockDim.x) header[i] += base32;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (base_out) *base_out = base32;
        if (overflow) atomicOr(err, err_offset_overflow_bit);
    }
}

template<bool Pack>
hipError_t launch_border(int dtype, void *data, void *body, const uint32_t *header, uint32_t nhc, const uint32_t *header_base,
        const border_geom &bg, uint32_t *out_len, uint32_t len_extra, hipStream_t stream, uint32_t *err, uint32_t body_words) {
    if (bg.count == 0) {
        if (Pack && out_len && nhc == 0) hipLaunchKernelGGL(store_length_kernel, dim3(1), dim3(1), 0, stream, out_len, len_extra);
        return hipGetLastError();
    }
    uint64_t blocks = (bg.count + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    if (dtype == NDZIP_HIP_F32) {
        hipLaunchKernelGGL((border_kernel<uint32_t, Pack>), dim3(static_cast<uint32_t>(blocks)), dim3(256), 0, stream,
                static_cast<uint32_t *>(data), static_cast<uint32_t *>(body), header, nhc, header_base, bg, out_len, len_extra, err, body_words);
    } else {
        hipLaunchKernelGGL((border_kernel<uint64_t, Pack>), dim3(static_cast<uint32_t>(blocks)), dim3(256), 0, stream,
                static_cast<uint64_t *>(data), static_cast<uint64_t *>(body), header, nhc, header_base, bg, out_len, len_extra, err, body_words);
    }
    return hipGetLastError();
}

// error word bits
constexpr uint32_t err_lookback_timeout = 1u, err_corrupt_header = 2u, err_offset_overflow = err_offset_overflow_bit;

int check_error_word(uint32_t *d_err, hipStream_t stream, uint32_t *bits = nullptr) {
    uint32_t host = 0;
    if (bits) *bits = 0;
    HIP_TRY(hipMemcpyAsync(&host, d_err, sizeof host, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    if (bits) *bits = host;
    if (host != 0) {
        HIP_TRY(hipMemsetAsync(d_err, 0, sizeof host, stream));
        char buf[224];
        snprintf(buf, sizeof buf, "device error word 0x%x (%s%s%s)", host, (host & err_lookback_timeout) ? "scan look-back timeout " : "",
                (host & err_corrupt_header) ? "corrupt stream header " : "",
                (host & err_offset_overflow) ? "sharded stream exceeds the format's 32-bit offsets" : "");
        return fail(NDZIP_HIP_ERR_DEVICE_FAULT, buf);
    }
    return NDZIP_HIP_OK;
}

}  // namespace

struct ndzip_hip_compressor {
    int dtype;
    int dims;
    uint32_t max_nhc;
    hipStream_t stream;
    tile_desc *desc;
    uint32_t *err;
    int num_cus;
    int device;          // the device the handle was created on (and launches on)
    int max_blocks_per_cu;  // 0 = full persistent grid; set to 1 for the relaunch after a look-back time-out (host-pointer paths)
    size_t desc_count;   // entries of `desc`
};

struct ndzip_hip_decompressor {
    int dtype;
    int dims;
    hipStream_t stream;
    uint32_t *err;
    int num_xcds;
    int f64_work_items = 0;  // 0 = default mapping of the 64-bit decoder (default_f64_work_items, codec_launch.hpp); 128 / 256 = an explicit choice
};

extern "C" {

const char *ndzip_hip_last_error(void) { return g_last_error.c_str(); }

int ndzip_hip_abi_version(void) { return NDZIP_HIP_ABI_VERSION; }

int ndzip_hip_device_info(char *arch, size_t arch_capacity, int *num_compute_units) {
    int cus = 0;
    if (int s = ensure_device(&cus)) return s;
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, dev));
    if (arch && arch_capacity) {
        std::string name = prop.gcnArchName;
        const auto colon = name.find(':');
        if (colon != std::string::npos) name.resize(colon);
        snprintf(arch, arch_capacity, "%s", name.c_str());
    }
    if (num_compute_units) *num_compute_units = cus;
    return NDZIP_HIP_OK;
}

int ndzip_hip_compressed_length_bound(int dtype, int dims, const uint32_t *extent, uint64_t *words) {
    if (!valid_dtype(dtype) || !valid_dims(dims) || !extent || !words) return fail(NDZIP_HIP_ERR_INVALID_ARGUMENT, "invalid argument");
    const grid_geom gg = make_geom(dims, extent);
    if (num_elements(gg) > 0xffffffffull) return fail(NDZIP_HIP_ERR_LIMIT, "extent has more than 2^32-1 elements (index_type is uint32_t)");
    *words = length_bound(dtype, gg);
    return NDZIP_HIP_OK;
}

int ndzip_hip_num_hypercubes(int dims, const uint32_t *extent, uint32_t *num_hypercubes) {
    if (!valid_dims(dims) || !extent || !num_hypercubes) return fail(NDZIP_HIP_ERR_INVALID_ARGUMENT, "invalid argument");
    const grid_geom gg = make_geom(dims, extent);
    if (num_elements(gg) > 0xffffffffull) return fail(NDZIP_HIP_ERR_LIMIT, "extent has more than 2^32-1 elements (index_type is uint32_t)");
    *num_hypercubes = gg.nhc;
    return NDZIP_HIP_OK;
}

int ndzip_hip_header_words(int dtype, uint32_t num_hypercubes, uint32_t *words) {
    if (!valid_dtype(dtype) || !words) return fail(NDZIP_HIP_ERR_INVALID_ARGUMENT, "invalid argument");
    *words = header_words_for(dtype, num_hypercubes);
    return NDZIP_HIP_OK;
}

int ndzip_hip_compressor_create(int dtype, int dims, uint32_t max_num_hypercubes, void *hip_stream, ndzip_hip_compressor **out) {
    if (!out) return fail(NDZIP_HIP_ERR_INVALID_ARGUMENT, "null handle pointer");
    *out = nullptr;
    if (!valid_dtype(dtype)) return fail(NDZIP_HIP_ERR_INVALID_ARGUMENT, "invalid dtype");
    if (!valid_dims(dims)) return fail(NDZIP_HIP_ERR_INVALID_ARGUMENT, "Invalid dimensionality");  // common.hh:642
    int cus = 0, device = 0;
    if (int s = ensure_device(&cus, &device)) return s;
    auto *c = new ndzip_hip_compressor{dtype, dims, max_num_hypercubes, static_cast<hipStream_t>(hip_stream), nullptr, nullptr, cus, device, 0, 0};
    const uint32_t tiles = dtype == NDZIP_HIP_F32 ? compress_num_tiles<float>(dims, max_num_hypercubes)
                                                  : compress_num_tiles<double>(dims, max_num_hypercubes);
    c->desc_count = static_cast<size_t>(tiles) + scratch_extra_descs;
    hipError_t e = hipMalloc(reinterpret_cast<void **>(&c->desc), c->desc_count * sizeof(tile_desc));
    // zeroed once: ticket counters start at 0 (the kernel restores that on its way out), descriptors carry the launch epoch, which
    // lives in the scratch and is advanced by the kernels themselves -- a compress call has no per-launch state on the host and can
    // be recorded into a hipGraph
    if (e == hipSuccess) e = hipMemsetAsync(c->desc, 0, c->desc_count * sizeof(tile_desc), c->stream);
    if (e == hipSuccess) e = init_scratch_epoch(c->desc, tiles, c->stream);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&c->err), sizeof(uint32_t));
    if (e == hipSuccess) e = hipMemsetAsync(c->err, 0, sizeof(uint32_t), c->stream);
    if (e != hipSuccess) {
        if (c->desc) (void) hipFree(c->desc);
        if (c->err) (void) hipFree(c->err);
        delete c;
        return fail_hip(e, "allocating compressor scratch");
    }
    *out = c;
    return NDZIP_HIP_OK;
}

static int compress_common(ndzip_hip_compressor *c, const void *d_in, int dims, const uint32_t *extent, uint32_t *d_header,
        void *d_body, uint32_t *d_len, bool split) {
    if (!c || !extent || !d_header || !d_body) return fail(NDZIP_HIP_ERR_INVALID_ARGUMENT, "null argument");
    if (dims != c->dims) return fail(NDZIP_HIP_ERR_DIMS_MISMATCH, "data dimensionality does not match compressor dimensionality");
    const grid_geom gg = make_geom(dims, extent);
    if (int s = check_limits(c->dtype, gg)) return s;
    if (gg.nhc > c->max_nhc) return fail(NDZIP_HIP_ERR_CAPACITY, "extent has more hypercubes than the compressor was created for");
    if (!d_in && num_elements(gg) > 0) return fail(NDZIP_HIP_ERR_INVALID_ARGUMENT, "null input");
    const border_geom bg = make_border_geom(gg);
    const uint32_t hw = header_words_for(c->dtype, gg.nhc);
    const uint32_t len_extra = (split ? 0u : hw) + static_cast<uint32_t>(bg.count);

    compress_args a{};
    a.in = d_in;
    a.gg = gg;
    a.header = d_header;
    a.body = d_body;
    a.desc = c->desc;
    a.out_len = d_len;
    a.len_extra = len_extra;
    a.err = c->err;
    a.stream = c->stream;
    a.num_cus = c->num_cus;
    a.device = c->device;
    a.max_blocks_per_cu = c->max_blocks_per_cu;
    a.aligned = is_aligned(c->dtype, gg, d_in);
    if (verbose()) fprintf(stderr, "[ndzip-hip] compress: %u hypercubes, %llu border elements\n", gg.nhc,
            static_cast<unsigned long long>(bg.count));
    if (gg.nhc > 0) {
        HIP_TRY(c->dtype == NDZIP_HIP_F32 ? launch_compress<float>(dims, a) : launch_compress<double>(dims, a));
    }
    HIP_TRY(launch_border<true>(c->dtype, const_cast<void *>(d_in), d_body, d_header, gg.nhc, nullptr, bg, d_len, len_extra, c->stream,
            c->err, 0xffffffffu));
    return NDZIP_HIP_OK;
}

int ndzip_hip_compressor_compress(ndzip_hip_compressor *c, const void *d_in, int dims, const uint32_t *extent, void *d_stream,
        uint32_t *d_stream_length_words) {
    if (!c || !extent || !d_stream) return fail(NDZIP_HIP_ERR_INVALID_ARGUMENT, "null argument");
    if (!valid_dims(dims)) return fail(NDZIP_HIP_ERR_INVALID_ARGUMENT, "Invalid dimensionality");
    const uint32_t nhc = make_geom(dims, extent).nhc;
    const uint32_t hw = header_words_for(c->dtype, nhc);
    void *body = static_cast<char *>(d_stream) + static_cast<size_t>(hw) * word_bytes(c->dtype);
    return compress_common(c, d_in, dims, extent, static_cast<uint32_t *>(d_stream), body, d_stream_length_words, false);
}

int ndzip_hip_compressor_compress_split(ndzip_hip_compressor *c, const void *d_in, int dims, const uint32_t *extent,
        uint32_t *d_header, void *d_body, uint32_t *d_body_length_words) {
    if (!valid_dims(dims)) return fail(NDZIP_HIP_ERR_INVALID_ARGUMENT, "Invalid dimensionality");
    return compress_common(c, d_in, dims, extent, d_header, d_body, d_body_length_words, true);
}

int ndzip_hip_compressor_offset_header(ndzip_hip_compressor *c, uint32_t *d_header, uint32_t count, uint32_t base) {
    if (!c || (!d_header && count)) return fail(NDZIP_HIP_ERR_INVALID_ARGUMENT, "null argument");
    if (count == 0 || base == 0) return NDZIP_HIP_OK;
    uint32_t blocks = (count + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(offset_header_kernel, dim3(blocks), dim3(256), 0, c->stream, d_header, count, base);
    HIP_TRY(hipGetLastError());
    return NDZIP_HIP_OK;
}

int ndzip_hip_compressor_offset_header_device(ndzip_hip_compressor *c, uint32_t *d_header, uint32_t count, const uint32_t *d_base) {
    if (!c || (!d_header && count) || !d_base) return fail(NDZIP_HIP_ERR_INVALID_ARGUMENT, "null argument");
    if (count == 0) return NDZIP_HIP_OK;
    uint32_t blocks = (count + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(offset_header_device_kernel, dim3(blocks), dim3(256), 0, c->stream, d_header, count, d_base);
    HIP_TRY(hipGetLastError());
    return NDZIP_HIP_OK;
}

int ndzip_hip_compressor_offset_header_gathered(ndzip_hip_compressor *c, uint32_t *d_header, uint32_t count, const uint32_t *d_lengths,
        const uint32_t *d_borders, uint32_t rank, uint32_t world, uint32_t *d_base_out) {
    if (!c || (!d_header && count) || !d_lengths || !d_borders) return fail(NDZIP_HIP_ERR_INVALID_ARGUMENT, "null argument");
    if (rank >= world) return fail(NDZIP_HIP_ERR_INVALID_ARGUMENT, "rank outside the plan");
    uint32_t blocks = (count + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (blocks == 0) blocks = 1;  // still publishes the base
    hipLaunchKernelGGL(offset_header_gathered_kernel, dim3(blocks), dim3(256), 0, c->stream, d_header, count, d_lengths, d_borders, rank,
            world, d_base_out, c->err);
    HIP_TRY(hipGetLastError());
    return NDZIP_HIP_OK;
}

int ndzip_hip_compressor_check(ndzip_hip_compressor *c) {
    if (!c) return fail(NDZIP_HIP_ERR_INVALID_ARGUMENT, "null handle");
    return check_error_word(c->err, c->stream);
}

int ndzip_hip_compressor_set_max_workgroups_per_cu(ndzip_hip_compressor *c, int max_workgroups_per_cu) {
    if (!c) return fail(NDZIP_HIP_ERR_INVALID_ARGUMENT, "null handle");
    if (max_workgroups_per_cu < 0 || max_workgroups_per_cu > 64) return fail(NDZIP_HIP_ERR_INVALID_ARGUMENT, "workgroups per CU: 0 (default) .. 64");
    c->max_blocks_per_cu = max_workgroups_per_cu;
    return NDZIP_HIP_OK;
}

int ndzip_hip_compressor_destroy(ndzip_hip_compressor *c) {
    if (!c) return NDZIP_HIP_OK;
    (void) hipStreamSynchronize(c->stream);
    (void) hipFree(c->desc);
    (void) hipFree(c->err);
    delete c;
    return NDZIP_HIP_OK;
}

int ndzip_hip_decompressor_create(int dtype, int dims, void *hip_stream, ndzip_hip_decompressor **out) {
    if (!out) return fail(NDZIP_HIP_ERR_INVALID_ARGUMENT, "null handle pointer");
    *out = nullptr;
    if (!valid_dtype(dtype)) return fail(NDZIP_HIP_ERR_INVALID_ARGUMENT, "invalid dtype");
    if (!valid_dims(dims)) return fail(NDZIP_HIP_ERR_INVALID_ARGUMENT, "Invalid dimensionality");
    int xcds = 1;
    if (int s = ensure_device(nullptr, nullptr, &xcds)) return s;
    auto *d = new ndzip_hip_decompressor{dtype, dims, static_cast<hipStream_t>(hip_stream), nullptr, xcds};
    hipError_t e = hipMalloc(reinterpret_cast<void **>(&d->err), sizeof(uint32_t));
    if (e == hipSuccess) e = hipMemsetAsync(d->err, 0, sizeof(uint32_t), d->stream);
    if (e != hipSuccess) {
        if (d->err) (void) hipFree(d->err);
        delete d;
        return fail_hip(e, "allocating decompressor scratch");
    }
    *out = d;
    return NDZIP_HIP_OK;
}

static int decompress_common(ndzip_hip_decompressor *d, const uint32_t *d_header, const uint32_t *header_base, const void *d_body,
        uint32_t body_words, void *d_out, int dims, const uint32_t *extent) {
    if (!d || !extent || !d_body) return fail(NDZIP_HIP_ERR_INVALID_ARGUMENT, "null argument");
    if (!valid_dims(dims)) return fail(NDZIP_HIP_ERR_INVALID_ARGUMENT, "Invalid dimensionality");
    if (dims != d->dims) return fail(NDZIP_HIP_ERR_DIMS_MISMATCH, "data dimensionality does not match decompressor dimensionality");
    const grid_geom gg = make_geom(dims, extent);
    if (int s = check_limits(d->dtype, gg)) return s;
    if (!d_out && num_elements(gg) > 0) return fail(NDZIP_HIP_ERR_INVALID_ARGUMENT, "null output");
    if (!d_header && gg.nhc > 0) return fail(NDZIP_HIP_ERR_INVALID_ARGUMENT, "null header");
    decompress_args a{};
    a.header = d_header;
    a.header_base = header_base;
    a.body = d_body;
    a.out = d_out;
    a.gg = gg;
    a.err = d->err;
    a.stream = d->stream;
    a.aligned = is_aligned(d->dtype, gg, d_out);
    a.body_words = body_words;
    a.num_xcds = d->num_xcds;
    a.f64_work_items = d->f64_work_items;
    if (verbose()) fprintf(stderr, "[ndzip-hip] decompress: %u hypercubes, %llu border elements\n", gg.nhc,
            static_cast<unsigned long long>(border_count(gg)));
    if (gg.nhc > 0) {
        HIP_TRY(d->dtype == NDZIP_HIP_F32 ? launch_decompress<float>(dims, a) : launch_decompress<double>(dims, a));
    }
    const border_geom bg = make_border_geom(gg);
    HIP_TRY(launch_border<false>(d->dtype, d_out, const_cast<void *>(d_body), d_header, gg.nhc, header_base, bg, nullptr, 0, d->stream,
            d->err, body_words));
    return NDZIP_HIP_OK;
}

int ndzip_hip_decompressor_decompress_split(ndzip_hip_decompressor *d, const uint32_t *d_header, const uint32_t *header_base,
        const void *d_body, void *d_out, int dims, const uint32_t *extent) {
    return decompress_common(d, d_header, header_base, d_body, 0xffffffffu, d_out, dims, extent);
}

int ndzip_hip_decompressor_decompress_split_bounded(ndzip_hip_decompressor *d, const uint32_t *d_header, const uint32_t *header_base,
        const void *d_body, uint32_t body_words, void *d_out, int dims, const uint32_t *extent) {
    return decompress_common(d, d_header, header_base, d_body, body_words, d_out, dims, extent);
}

int ndzip_hip_decompressor_decompress(ndzip_hip_decompressor *d, const void *d_stream, void *d_out, int dims, const uint32_t *extent) {
    return ndzip_hip_decompressor_decompress_bounded(d, d_stream, 0xffffffffu, d_out, dims, extent);
}

int ndzip_hip_decompressor_decompress_bounded(ndzip_hip_decompressor *d, const void *d_stream, uint32_t stream_length_words, void *d_out,
        int dims, const uint32_t *extent) {
    if (!d || !extent || !d_stream) return fail(NDZIP_HIP_ERR_INVALID_ARGUMENT, "null argument");
    if (!valid_dims(dims)) return fail(NDZIP_HIP_ERR_INVALID_ARGUMENT, "Invalid dimensionality");
    const uint32_t nhc = make_geom(dims, extent).nhc;
    const uint32_t hw = header_words_for(d->dtype, nhc);
    if (stream_length_words < hw) return fail(NDZIP_HIP_ERR_INVALID_ARGUMENT, "stream shorter than its header");
    const void *body = static_cast<const char *>(d_stream) + static_cast<size_t>(hw) * word_bytes(d->dtype);
    const uint32_t body_words = stream_length_words == 0xffffffffu ? 0xffffffffu : stream_length_words - hw;
    return decompress_common(d, static_cast<const uint32_t *>(d_stream), nullptr, body, body_words, d_out, dims, extent);
}

int ndzip_hip_decompressor_set_f64_work_items(ndzip_hip_decompressor *d, int work_items_per_hypercube) {
    if (!d) return fail(NDZIP_HIP_ERR_INVALID_ARGUMENT, "null handle");
    if (work_items_per_hypercube != 0 && work_items_per_hypercube != 128 && work_items_per_hypercube != 256) {
        return fail(NDZIP_HIP_ERR_INVALID_ARGUMENT, "work-items per 64-bit hypercube: 0 (default), 128 or 256");
    }
    d->f64_work_items = work_items_per_hypercube;
    return NDZIP_HIP_OK;
}

int ndzip_hip_decompressor_check(ndzip_hip_decompressor *d) {
    if (!d) return fail(NDZIP_HIP_ERR_INVALID_ARGUMENT, "null handle");
    return check_error_word(d->err, d->stream);
}

int ndzip_hip_decompressor_destroy(ndzip_hip_decompressor *d) {
    if (!d) return NDZIP_HIP_OK;
    (void) hipStreamSynchronize(d->stream);
    (void) hipFree(d->err);
    delete d;
    return NDZIP_HIP_OK;
}

// ---- host-pointer interface (cuda_offloader, cuda_codec.inl:654-761) ---------------------------------------------

namespace {
struct device_buffer {
    void *p = nullptr;
    ~device_buffer() {
        if (p) (void) hipFree(p);
    }
    hipError_t allocate(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 16); }
};
struct event_pair {
    hipEvent_t start = nullptr, stop = nullptr;
    ~event_pair() {
        if (start) (void) hipEventDestroy(start);
        if (stop) (void) hipEventDestroy(stop);
    }
};
}  // namespace

int ndzip_hip_offload_compress(int dtype, int dims, const uint32_t *extent, const void *data, void *stream,
        uint32_t *stream_length_words, uint64_t *kernel_ns) {
    if (!valid_dtype(dtype) || !extent || !stream || !stream_length_words) return fail(NDZIP_HIP_ERR_INVALID_ARGUMENT, "invalid argument");
    if (!valid_dims(dims)) return fail(NDZIP_HIP_ERR_INVALID_ARGUMENT, "Invalid dimensionality");
    if (int s = ensure_device(nullptr)) return s;
    const grid_geom gg = make_geom(dims, extent);
    if (int s = check_limits(dtype, gg)) return s;
    const size_t wb = word_bytes(dtype);
    const size_t in_bytes = num_elements(gg) * wb;
    const size_t bound_bytes = length_bound(dtype, gg) * wb;
    device_buffer d_in, d_stream, d_len;
    HIP_TRY(d_in.allocate(in_bytes));
    HIP_TRY(d_stream.allocate(bound_bytes));
    HIP_TRY(d_len.allocate(sizeof(uint32_t)));
    if (in_bytes) HIP_TRY(hipMemcpy(d_in.p, data, in_bytes, hipMemcpyHostToDevice));
    ndzip_hip_compressor *c = nullptr;
    if (int s = ndzip_hip_compressor_create(dtype, dims, gg.nhc, nullptr, &c)) return s;
    event_pair ev;
    const bool timed = kernel_ns != nullptr || verbose();
    int status = NDZIP_HIP_OK;
    // A look-back that timed out (a predecessor tile's workgroup did not publish within the poll budget: the persistent grid
    // was not fully resident, e.g. because another process held part of the GPU) spoils this launch only: the input is still on
    // the device, so the launch is repeated ONCE -- with one workgroup per CU instead of four, a grid that is far more likely
    // to be resident in full -- before the caller sees NDZIP_HIP_ERR_DEVICE_FAULT.  (The device-pointer entry points cannot do
    // that -- they never synchronise; their callers see the fault in ndzip_hip_compressor_check().)
    for (int attempt = 0;; ++attempt) {
        if (timed) {
            if (!ev.start && (hipEventCreate(&ev.start) != hipSuccess || hipEventCreate(&ev.stop) != hipSuccess)) {
                status = fail(NDZIP_HIP_ERR_RUNTIME, "hipEventCreate failed");
                break;
            }
            (void) hipEventRecord(ev.start, nullptr);
        }
        status = ndzip_hip_compressor_compress(c, d_in.p, dims, extent, d_stream.p, static_cast<uint32_t *>(d_len.p));
        if (status) break;
        if (timed) {
            (void) hipEventRecord(ev.stop, nullptr);
            (void) hipEventSynchronize(ev.stop);
            float ms = 0;
            (void) hipEventElapsedTime(&ms, ev.start, ev.stop);
            if (kernel_ns) *kernel_ns = static_cast<uint64_t>(static_cast<double>(ms) * 1e6);
            if (verbose()) fprintf(stderr, "[ndzip-hip][profile] total kernel time %.3fms\n", static_cast<double>(ms));
        }
        uint32_t bits = 0;
        status = check_error_word(c->err, c->stream, &bits);
        if (status == NDZIP_HIP_ERR_DEVICE_FAULT && bits == err_lookback_timeout && attempt == 0) {
            if (verbose()) fprintf(stderr, "[ndzip-hip] scan look-back timeout: relaunching once\n");
            c->max_blocks_per_cu = 1;  // (a quarter of the grid: what a GPU shared with another process is far more likely to hold in full)
            continue;                  // (the handle is private to this call and destroyed below)
        }
        if (status == NDZIP_HIP_ERR_DEVICE_FAULT && attempt > 0) status = fail(status, g_last_error + " -- again after one relaunch");
        break;
    }
    ndzip_hip_compressor_destroy(c);
    if (status) return status;
    uint32_t len = 0;
    HIP_TRY(hipMemcpy(&len, d_len.p, sizeof len, hipMemcpyDeviceToHost));
    if (static_cast<size_t>(len) * wb > bound_bytes) return fail(NDZIP_HIP_ERR_DEVICE_FAULT, "stream length exceeds bound");
    if (len) HIP_TRY(hipMemcpy(stream, d_stream.p, static_cast<size_t>(len) * wb, hipMemcpyDeviceToHost));
    *stream_length_words = len;
    return NDZIP_HIP_OK;
}

int ndzip_hip_offload_decompress(int dtype, int dims, const uint32_t *extent, const void *stream, uint32_t stream_length_words,
        void *data, uint32_t *words_consumed, uint64_t *kernel_ns) {
    if (!valid_dtype(dtype) || !extent || !stream) return fail(NDZIP_HIP_ERR_INVALID_ARGUMENT, "invalid argument");
    if (!valid_dims(dims)) return fail(NDZIP_HIP_ERR_INVALID_ARGUMENT, "Invalid dimensionality");
    if (int s = ensure_device(nullptr)) return s;
    const grid_geom gg = make_geom(dims, extent);
    if (int s = check_limits(dtype, gg)) return s;
    const size_t wb = word_bytes(dtype);
    const size_t out_bytes = num_elements(gg) * wb;
    // The reference trusts `length` and the header (cuda_codec.inl:726-730); here the header is checked on the host, where
    // it already is, so a truncated or corrupt stream is an error code instead of a device fault.
    uint32_t total_words = 0;
    if (int s = ndzip_hip_stream_words(dtype, dims, extent, stream, stream_length_words, &total_words)) return s;
    device_buffer d_stream, d_out;
    HIP_TRY(d_stream.allocate(static_cast<size_t>(total_words) * wb));
    HIP_TRY(d_out.allocate(out_bytes));
    if (total_words) HIP_TRY(hipMemcpy(d_stream.p, stream, static_cast<size_t>(total_words) * wb, hipMemcpyHostToDevice));
    ndzip_hip_decompressor *d = nullptr;
    if (int s = ndzip_hip_decompressor_create(dtype, dims, nullptr, &d)) return s;
    event_pair ev;
    const bool timed = kernel_ns != nullptr || verbose();
    int status = NDZIP_HIP_OK;
    do {
        if (timed) {
            if (hipEventCreate(&ev.start) != hipSuccess || hipEventCreate(&ev.stop) != hipSuccess) {
                status = fail(NDZIP_HIP_ERR_RUNTIME, "hipEventCreate failed");
                break;
            }
            (void) hipEventRecord(ev.start, nullptr);
        }
        status = ndzip_hip_decompressor_decompress_bounded(d, d_stream.p, total_words, d_out.p, dims, extent);
        if (status) break;
        if (timed) {
            (void) hipEventRecord(ev.stop, nullptr);
            (void) hipEventSynchronize(ev.stop);
            float ms = 0;
            (void) hipEventElapsedTime(&ms, ev.start, ev.stop);
            if (kernel_ns) *kernel_ns = static_cast<uint64_t>(static_cast<double>(ms) * 1e6);
            if (verbose()) fprintf(stderr, "[ndzip-hip][profile] total kernel time %.3fms\n", static_cast<double>(ms));
        }
        status = ndzip_hip_decompressor_check(d);
    } while (false);
    ndzip_hip_decompressor_destroy(d);
    if (status) return status;
    if (out_bytes) HIP_TRY(hipMemcpy(data, d_out.p, out_bytes, hipMemcpyDeviceToHost));
    // border offset + border words, recomputed from the header on the host (cuda_codec.inl:740-745)
    if (words_consumed) *words_consumed = total_words;
    return NDZIP_HIP_OK;
}

// ---- persistent, pipelined host-pointer interface ---------------------------------------------------------------

}  // extern "C"

namespace {
struct offload_slot {
    hipStream_t stream = nullptr;
    void *d_array = nullptr;    // raw array (input of compress, output of decompress)
    void *d_stream = nullptr;   // compressed stream, length-bound words
    uint32_t *d_len = nullptr;
    uint32_t *h_len = nullptr;  // pinned
    hipEvent_t start = nullptr, stop = nullptr;
    ndzip_hip_compressor *comp = nullptr;
    ndzip_hip_decompressor *decomp = nullptr;
    int job = 0;  // 0 idle, 1 compress, 2 decompress
    void *host_out = nullptr;
    uint32_t words = 0;  // decompress: words consumed, known at submit
    uint32_t extent[3] = {0, 0, 0};  // compress: the job's extent (for the relaunch after a look-back timeout)
};
}  // namespace

struct ndzip_hip_offloader {
    int dtype;
    int dims;
    grid_geom max_gg;
    size_t array_bytes;
    size_t stream_bytes;
    int nslots;
    offload_slot *slots;
};

namespace {
void destroy_offloader(ndzip_hip_offloader *o) {
    if (!o) return;
    for (int i = 0; i < o->nslots; ++i) {
        offload_slot &s = o->slots[i];
        if (s.stream) (void) hipStreamSynchronize(s.stream);
        if (s.comp) ndzip_hip_compressor_destroy(s.comp);
        if (s.decomp) ndzip_hip_decompressor_destroy(s.decomp);
        if (s.d_array) (void) hipFree(s.d_array);
        if (s.d_stream) (void) hipFree(s.d_stream);
        if (s.d_len) (void) hipFree(s.d_len);
        if (s.h_len) (void) hipHostFree(s.h_len);
        if (s.start) (void) hipEventDestroy(s.start);
        if (s.stop) (void) hipEventDestroy(s.stop);
        if (s.stream) (void) hipStreamDestroy(s.stream);
    }
    delete[] o->slots;
    delete o;
}

int slot_of(ndzip_hip_offloader *o, int slot, offload_slot **out, bool want_idle) {
    if (!o) return fail(NDZIP_HIP_ERR_INVALID_ARGUMENT, "null handle");
    if (slot < 0 || slot >= o->nslots) return fail(NDZIP_HIP_ERR_INVALID_ARGUMENT, "slot out of range");
    if (want_idle && o->slots[slot].job != 0) return fail(NDZIP_HIP_ERR_INVALID_ARGUMENT, "slot is busy: wait for it first");
    if (!want_idle && o->slots[slot].job == 0) return fail(NDZIP_HIP_ERR_INVALID_ARGUMENT, "slot is idle: nothing to wait for");
    *out = &o->slots[slot];
    return NDZIP_HIP_OK;
}

int job_geometry(ndzip_hip_offloader *o, const uint32_t *extent, grid_geom *gg) {
    if (!extent) return fail(NDZIP_HIP_ERR_INVALID_ARGUMENT, "null extent");
    *gg = make_geom(o->dims, extent);
    if (int s = check_limits(o->dtype, *gg)) return s;
    if (num_elements(*gg) * word_bytes(o->dtype) > o->array_bytes || length_bound(o->dtype, *gg) * word_bytes(o->dtype) > o->stream_bytes
            || gg->nhc > o->max_gg.nhc) {
        return fail(NDZIP_HIP_ERR_CAPACITY, "extent is larger than the offloader was created for");
    }
    return NDZIP_HIP_OK;
}
}  // namespace

extern "C" {

int ndzip_hip_offloader_create(int dtype, int dims, const uint32_t *max_extent, int slots, ndzip_hip_offloader **out) {
    if (!out) return fail(NDZIP_HIP_ERR_INVALID_ARGUMENT, "null handle pointer");
    *out = nullptr;
    if (!valid_dtype(dtype) || !max_extent) return fail(NDZIP_HIP_ERR_INVALID_ARGUMENT, "invalid argument");
    if (!valid_dims(dims)) return fail(NDZIP_HIP_ERR_INVALID_ARGUMENT, "Invalid dimensionality");
    if (slots < 1 || slots > 16) return fail(NDZIP_HIP_ERR_INVALID_ARGUMENT, "slots must be between 1 and 16");
    if (int s = ensure_device(nullptr)) return s;
    const grid_geom gg = make_geom(dims, max_extent);
    if (int s = check_limits(dtype, gg)) return s;
    auto *o = new ndzip_hip_offloader{dtype, dims, gg, static_cast<size_t>(num_elements(gg)) * word_bytes(dtype),
            static_cast<size_t>(length_bound(dtype, gg)) * word_bytes(dtype), slots, new offload_slot[slots]};
    for (int i = 0; i < slots; ++i) {
        offload_slot &s = o->slots[i];
        hipError_t e = hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipMalloc(&s.d_array, o->array_bytes ? o->array_bytes : 16);
        if (e == hipSuccess) e = hipMalloc(&s.d_stream, o->stream_bytes ? o->stream_bytes : 16);
        if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&s.d_len), sizeof(uint32_t));
        if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void **>(&s.h_len), sizeof(uint32_t), hipHostMallocDefault);
        if (e == hipSuccess) e = hipEventCreate(&s.start);
        if (e == hipSuccess) e = hipEventCreate(&s.stop);
        int status = e == hipSuccess ? NDZIP_HIP_OK : fail_hip(e, "allocating offloader slot");
        if (!status) status = ndzip_hip_compressor_create(dtype, dims, gg.nhc, s.stream, &s.comp);
        if (!status) status = ndzip_hip_decompressor_create(dtype, dims, s.stream, &s.decomp);
        if (status) {
            destroy_offloader(o);
            return status;
        }
    }
    *out = o;
    return NDZIP_HIP_OK;
}

int ndzip_hip_offloader_destroy(ndzip_hip_offloader *o) {
    destroy_offloader(o);
    return NDZIP_HIP_OK;
}

int ndzip_hip_host_alloc(size_t bytes, void **ptr) {
    if (!ptr) return fail(NDZIP_HIP_ERR_INVALID_ARGUMENT, "null pointer");
    *ptr = nullptr;
    if (int s = ensure_device(nullptr)) return s;
    HIP_TRY(hipHostMalloc(ptr, bytes ? bytes : 16, hipHostMallocDefault));
    return NDZIP_HIP_OK;
}

int ndzip_hip_host_free(void *ptr) {
    if (ptr) HIP_TRY(hipHostFree(ptr));
    return NDZIP_HIP_OK;
}

int ndzip_hip_offloader_submit_compress(ndzip_hip_offloader *o, int slot, const uint32_t *extent, const void *data, void *stream) {
    offload_slot *s = nullptr;
    if (int st = slot_of(o, slot, &s, true)) return st;
    if (!stream) return fail(NDZIP_HIP_ERR_INVALID_ARGUMENT, "null stream buffer");
    grid_geom gg;
    if (int st = job_geometry(o, extent, &gg)) return st;
    const size_t in_bytes = static_cast<size_t>(num_elements(gg)) * word_bytes(o->dtype);
    if (in_bytes && !data) return fail(NDZIP_HIP_ERR_INVALID_ARGUMENT, "null input");
    if (in_bytes) HIP_TRY(hipMemcpyAsync(s->d_array, data, in_bytes, hipMemcpyHostToDevice, s->stream));
    HIP_TRY(hipEventRecord(s->start, s->stream));
    if (int st = ndzip_hip_compressor_compress(s->comp, s->d_array, o->dims, extent, s->d_stream, s->d_len)) return st;
    HIP_TRY(hipEventRecord(s->stop, s->stream));
    HIP_TRY(hipMemcpyAsync(s->h_len, s->d_len, sizeof(uint32_t), hipMemcpyDeviceToHost, s->stream));
    s->job = 1;
    s->host_out = stream;
    for (int d = 0; d < 3; ++d) s->extent[d] = d < o->dims ? extent[d] : 1u;
    return NDZIP_HIP_OK;
}

int ndzip_hip_stream_words(int dtype, int dims, const uint32_t *extent, const void *stream, uint64_t available_words, uint32_t *words) {
    if (!valid_dtype(dtype) || !extent || !words) return fail(NDZIP_HIP_ERR_INVALID_ARGUMENT, "invalid argument");
    if (!valid_dims(dims)) return fail(NDZIP_HIP_ERR_INVALID_ARGUMENT, "Invalid dimensionality");
    const grid_geom gg = make_geom(dims, extent);
    if (num_elements(gg) > 0xffffffffull) return fail(NDZIP_HIP_ERR_LIMIT, "extent has more than 2^32-1 elements (index_type is uint32_t)");
    const uint64_t hw = header_words_for(dtype, gg.nhc);
    if (available_words < hw || (gg.nhc && !stream)) return fail(NDZIP_HIP_ERR_INVALID_ARGUMENT, "stream shorter than its header");
    // every entry: offset_after(hc) - offset_after(hc - 1) is the length of one encoded hypercube, head words (4096 / B)
    // up to head words + 4096 (common.hh:391-392) -- which also makes the entries strictly increasing
    const uint64_t B = dtype == NDZIP_HIP_F32 ? 32 : 64;
    const uint64_t min_len = hc_size / B, max_len = hc_size / B * (B + 1);
    const uint32_t *entries = static_cast<const uint32_t *>(stream);
    uint64_t last = 0;
    for (uint32_t hc = 0; hc < gg.nhc; ++hc) {
        const uint64_t e = entries[hc];
        if (e < last + min_len || e > last + max_len) {
            char buf[160];
            snprintf(buf, sizeof buf, "corrupt stream header: entry %u = %llu after %llu is not one encoded hypercube", hc,
                    static_cast<unsigned long long>(e), static_cast<unsigned long long>(last));
            return fail(NDZIP_HIP_ERR_INVALID_ARGUMENT, buf);
        }
        last = e;
    }
    const uint64_t total = hw + last + border_count(gg);
    if (total > 0xffffffffull) return fail(NDZIP_HIP_ERR_LIMIT, "stream length exceeds 2^32-1 words");
    if (total > available_words) return fail(NDZIP_HIP_ERR_INVALID_ARGUMENT, "header describes a stream longer than the given words");
    *words = static_cast<uint32_t>(total);
    return NDZIP_HIP_OK;
}

int ndzip_hip_offloader_submit_decompress(ndzip_hip_offloader *o, int slot, const uint32_t *extent, const void *stream,
        uint32_t stream_length_words, void *data) {
    offload_slot *s = nullptr;
    if (int st = slot_of(o, slot, &s, true)) return st;
    grid_geom gg;
    if (int st = job_geometry(o, extent, &gg)) return st;
    uint32_t words = 0;
    if (int st = ndzip_hip_stream_words(o->dtype, o->dims, extent, stream, stream_length_words, &words)) return st;
    const size_t wb = word_bytes(o->dtype);
    const size_t out_bytes = static_cast<size_t>(num_elements(gg)) * wb;
    if (out_bytes && !data) return fail(NDZIP_HIP_ERR_INVALID_ARGUMENT, "null output");
    if (words) HIP_TRY(hipMemcpyAsync(s->d_stream, stream, static_cast<size_t>(words) * wb, hipMemcpyHostToDevice, s->stream));
    HIP_TRY(hipEventRecord(s->start, s->stream));
    if (int st = ndzip_hip_decompressor_decompress_bounded(s->decomp, s->d_stream, words, s->d_array, o->dims, extent)) return st;
    HIP_TRY(hipEventRecord(s->stop, s->stream));
    if (out_bytes) HIP_TRY(hipMemcpyAsync(data, s->d_array, out_bytes, hipMemcpyDeviceToHost, s->stream));
    s->job = 2;
    s->host_out = data;
    s->words = words;
    return NDZIP_HIP_OK;
}

}  // extern "C"

namespace {
// `dest` != nullptr: a compress job's stream goes there (room for `dest_capacity_words`) instead of to the buffer named at
// submit time -- the copy happens here, when the exact length is known, so a caller that packs streams back to back
// (ndzip_hip_chunked_compress) names the final place only now.
// time between the slot's events, read once the job's (last) launch is known to have ended
int slot_kernel_time(offload_slot *s, int slot, uint64_t *kernel_ns) {
    if (!kernel_ns && !verbose()) return NDZIP_HIP_OK;
    HIP_TRY(hipEventSynchronize(s->stop));
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, s->start, s->stop));
    if (kernel_ns) *kernel_ns = static_cast<uint64_t>(static_cast<double>(ms) * 1e6);
    if (verbose()) fprintf(stderr, "[ndzip-hip][profile] slot %d total kernel time %.3fms\n", slot, static_cast<double>(ms));
    return NDZIP_HIP_OK;
}

int offloader_wait_impl(ndzip_hip_offloader *o, int slot, void *dest, uint64_t dest_capacity_words, uint32_t *words, uint64_t *kernel_ns) {
    offload_slot *s = nullptr;
    if (int st = slot_of(o, slot, &s, false)) return st;
    const int job = s->job;
    s->job = 0;
    HIP_TRY(hipStreamSynchronize(s->stream));
    if (job == 1) {
        uint32_t bits = 0;
        int st = check_error_word(s->comp->err, s->comp->stream, &bits);
        if (st == NDZIP_HIP_ERR_DEVICE_FAULT && bits == err_lookback_timeout) {
            // the array is still in the slot's device buffer: one relaunch (see ndzip_hip_offload_compress)
            if (verbose()) fprintf(stderr, "[ndzip-hip] slot %d: scan look-back timeout: relaunching once\n", slot);
            const int configured = s->comp->max_blocks_per_cu;
            s->comp->max_blocks_per_cu = 1;  // (one workgroup per CU for the retry; the slot goes back to its grid afterwards)
            // (the slot's events move to the relaunch: the time reported for the job is that of the launch whose stream is returned)
            HIP_TRY(hipEventRecord(s->start, s->stream));
            const int e = ndzip_hip_compressor_compress(s->comp, s->d_array, o->dims, s->extent, s->d_stream, s->d_len);
            s->comp->max_blocks_per_cu = configured;
            if (e) return e;
            HIP_TRY(hipEventRecord(s->stop, s->stream));
            HIP_TRY(hipMemcpyAsync(s->h_len, s->d_len, sizeof(uint32_t), hipMemcpyDeviceToHost, s->stream));
            st = check_error_word(s->comp->err, s->comp->stream, &bits);
            if (st == NDZIP_HIP_ERR_DEVICE_FAULT) st = fail(st, g_last_error + " -- again after one relaunch");
        }
        if (st) return st;
        if (int ts = slot_kernel_time(s, slot, kernel_ns)) return ts;
        const uint32_t len = *s->h_len;
        const size_t wb = word_bytes(o->dtype);
        if (static_cast<size_t>(len) * wb > o->stream_bytes) return fail(NDZIP_HIP_ERR_DEVICE_FAULT, "stream length exceeds bound");
        if (dest && len > dest_capacity_words) return fail(NDZIP_HIP_ERR_CAPACITY, "stream buffer too small for the compressed streams");
        if (len) {
            HIP_TRY(hipMemcpyAsync(dest ? dest : s->host_out, s->d_stream, static_cast<size_t>(len) * wb, hipMemcpyDeviceToHost, s->stream));
            HIP_TRY(hipStreamSynchronize(s->stream));
        }
        if (words) *words = len;
    } else {
        if (int st = ndzip_hip_decompressor_check(s->decomp)) return st;
        if (int ts = slot_kernel_time(s, slot, kernel_ns)) return ts;
        if (words) *words = s->words;
    }
    return NDZIP_HIP_OK;
}
}  // namespace

extern "C" {

int ndzip_hip_offloader_wait(ndzip_hip_offloader *o, int slot, uint32_t *words, uint64_t *kernel_ns) {
    return offloader_wait_impl(o, slot, nullptr, 0, words, kernel_ns);
}

// ---- arrays beyond the format's 32-bit counts: one stream per slab of dimension 0 -----------------------------------------

}  // extern "C"

namespace {
struct chunk_plan {
    uint64_t rows_per_chunk = 0;  // rows of dimension 0 per slab (a multiple of the hypercube side, except in a single-slab plan)
    uint64_t num_chunks = 0;
    uint64_t rest = 1;            // elements per row of dimension 0
};

// The fewest slabs of whole hypercube rows such that every slab is a legal ndzip array: fewer than `max_elements` elements
// (0 = the format's 2^32 - 1, ndzip.hh:20) and a length bound of at most 2^32 - 1 words.
int plan_chunks(int dtype, int dims, const uint64_t *extent, uint64_t max_elements, chunk_plan *plan) {
    if (!valid_dtype(dtype) || !extent || !plan) return fail(NDZIP_HIP_ERR_INVALID_ARGUMENT, "invalid argument");
    if (!valid_dims(dims)) return fail(NDZIP_HIP_ERR_INVALID_ARGUMENT, "Invalid dimensionality");
    const uint64_t limit = max_elements ? max_elements : 0xffffffffull;
    uint64_t rest = 1;
    for (int d = 1; d < dims; ++d) {
        if (extent[d] > 0xffffffffull || __builtin_mul_overflow(rest, extent[d], &rest)) return fail(NDZIP_HIP_ERR_LIMIT, "extent too large");
    }
    plan->rest = rest;
    const uint64_t side = side_for_dims(dims);
    uint64_t total = 0;
    const bool fits = !__builtin_mul_overflow(rest, extent[0], &total) && total <= limit && extent[0] <= 0xffffffffull;
    if (fits) {
        uint32_t e32[3] = {static_cast<uint32_t>(extent[0]), dims > 1 ? static_cast<uint32_t>(extent[1]) : 1u, dims > 2 ? static_cast<uint32_t>(extent[2]) : 1u};
        if (length_bound(dtype, make_geom(dims, e32)) <= 0xffffffffull) {
            plan->rows_per_chunk = extent[0];
            plan->num_chunks = 1;
            return NDZIP_HIP_OK;
        }
    }
    if (rest == 0 || extent[0] == 0) {
        plan->rows_per_chunk = extent[0];
        plan->num_chunks = 1;
        return NDZIP_HIP_OK;
    }
    // rows per slab: the largest multiple of the side whose elements and length bound fit (the bound is at most
    // elements * (B + 1) / B + header, so leave that margin), at least one row of hypercubes
    const uint64_t B = dtype == NDZIP_HIP_F32 ? 32 : 64;
    const uint64_t budget = limit < 0xffffffffull / (B + 2) * B ? limit : 0xffffffffull / (B + 2) * B;
    uint64_t rows = budget / rest / side * side;
    if (rows == 0) return fail(NDZIP_HIP_ERR_LIMIT, "one row of hypercubes along dimension 0 already exceeds the format's limits");
    const uint64_t n = (extent[0] + rows - 1) / rows;  // (the last slab takes what is left, border rows included)
    plan->rows_per_chunk = rows;
    plan->num_chunks = n;
    return NDZIP_HIP_OK;
}

void chunk_extent(const chunk_plan &p, int dims, const uint64_t *extent, uint64_t k, uint32_t out[3]) {
    const uint64_t r0 = k * p.rows_per_chunk;
    const uint64_t rows = extent[0] - r0 < p.rows_per_chunk ? extent[0] - r0 : p.rows_per_chunk;
    out[0] = static_cast<uint32_t>(rows);
    out[1] = dims > 1 ? static_cast<uint32_t>(extent[1]) : 1u;
    out[2] = dims > 2 ? static_cast<uint32_t>(extent[2]) : 1u;
}
}  // namespace

extern "C" {

int ndzip_hip_chunked_plan(int dtype, int dims, const uint64_t *extent, uint64_t max_elements, uint64_t *rows_per_chunk, uint64_t *num_chunks,
        uint64_t *length_bound_words) {
    chunk_plan p;
    if (int s = plan_chunks(dtype, dims, extent, max_elements, &p)) return s;
    if (rows_per_chunk) *rows_per_chunk = p.rows_per_chunk;
    if (num_chunks) *num_chunks = p.num_chunks;
    if (length_bound_words) {
        uint64_t total = 0;
        for (uint64_t k = 0; k < p.num_chunks; ++k) {
            uint32_t e[3];
            chunk_extent(p, dims, extent, k, e);
            total += length_bound(dtype, make_geom(dims, e));
        }
        *length_bound_words = total;
    }
    return NDZIP_HIP_OK;
}

int ndzip_hip_chunked_compress(int dtype, int dims, const uint64_t *extent, uint64_t max_elements, const void *data, void *streams,
        uint64_t capacity_words, uint64_t *total_words, uint64_t *kernel_ns) {
    if (!streams || !total_words) return fail(NDZIP_HIP_ERR_INVALID_ARGUMENT, "invalid argument");
    chunk_plan p;
    if (int s = plan_chunks(dtype, dims, extent, max_elements, &p)) return s;
    uint32_t first[3];
    chunk_extent(p, dims, extent, 0, first);
    ndzip_hip_offloader *o = nullptr;
    constexpr int slots = 2;
    if (int s = ndzip_hip_offloader_create(dtype, dims, first, slots, &o)) return s;
    const size_t wb = word_bytes(dtype);
    const char *in = static_cast<const char *>(data);
    char *out = static_cast<char *>(streams);
    uint64_t written = 0, ns_total = 0;
    // Slabs retire in order, and a slab's stream leaves the device only when it retires -- with its exact length known and its
    // predecessors' streams already in place: it is copied straight behind them (no provisional placement, no host-side move).
    int status = NDZIP_HIP_OK;
    auto retire = [&](uint64_t k) {
        uint32_t words = 0;
        uint64_t ns = 0;
        const int st = offloader_wait_impl(o, static_cast<int>(k % slots), out + written * wb, capacity_words - written, &words, &ns);
        if (st) return st;
        written += words;
        ns_total += ns;
        return static_cast<int>(NDZIP_HIP_OK);
    };
    for (uint64_t k = 0; k < p.num_chunks && !status; ++k) {
        if (k >= slots) status = retire(k - slots);
        if (status) break;
        uint32_t e[3];
        chunk_extent(p, dims, extent, k, e);
        status = ndzip_hip_offloader_submit_compress(o, static_cast<int>(k % slots), e, in + k * p.rows_per_chunk * p.rest * wb, out);
    }
    for (uint64_t k = p.num_chunks > slots ? p.num_chunks - slots : 0; k < p.num_chunks && !status; ++k) status = retire(k);
    ndzip_hip_offloader_destroy(o);
    if (status) return status;
    *total_words = written;
    if (kernel_ns) *kernel_ns = ns_total;
    return NDZIP_HIP_OK;
}

int ndzip_hip_chunked_decompress(int dtype, int dims, const uint64_t *extent, uint64_t max_elements, const void *streams, uint64_t total_words,
        void *data, uint64_t *words_consumed, uint64_t *kernel_ns) {
    if (!streams) return fail(NDZIP_HIP_ERR_INVALID_ARGUMENT, "invalid argument");
    chunk_plan p;
    if (int s = plan_chunks(dtype, dims, extent, max_elements, &p)) return s;
    uint32_t first[3];
    chunk_extent(p, dims, extent, 0, first);
    ndzip_hip_offloader *o = nullptr;
    constexpr int slots = 2;
    if (int s = ndzip_hip_offloader_create(dtype, dims, first, slots, &o)) return s;
    const size_t wb = word_bytes(dtype);
    const char *in = static_cast<const char *>(streams);
    char *out = static_cast<char *>(data);
    uint64_t consumed = 0, ns_total = 0;
    int status = NDZIP_HIP_OK;
    auto retire = [&](uint64_t k) {
        uint64_t ns = 0;
        const int st = ndzip_hip_offloader_wait(o, static_cast<int>(k % slots), nullptr, &ns);
        ns_total += ns;
        return st;
    };
    for (uint64_t k = 0; k < p.num_chunks && !status; ++k) {
        if (k >= slots) status = retire(k - slots);
        if (status) break;
        uint32_t e[3], words = 0;
        chunk_extent(p, dims, extent, k, e);
        status = ndzip_hip_stream_words(dtype, dims, e, in + consumed * wb, total_words - consumed, &words);
        if (status) break;
        status = ndzip_hip_offloader_submit_decompress(o, static_cast<int>(k % slots), e, in + consumed * wb, words, out + k * p.rows_per_chunk * p.rest * wb);
        consumed += words;
    }
    for (uint64_t k = p.num_chunks > slots ? p.num_chunks - slots : 0; k < p.num_chunks && !status; ++k) status = retire(k);
    ndzip_hip_offloader_destroy(o);
    if (status) return status;
    if (words_consumed) *words_consumed = consumed;
    if (kernel_ns) *kernel_ns = ns_total;
    return NDZIP_HIP_OK;
}

}  // extern "C"
