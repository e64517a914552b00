// include/ndzip_hip_sharded.hh -- header-only C++ adaptor over include/ndzip_hip_sharded.h (libndzip_hip_rccl.so) in the style of the
// reference's device-pointer classes (include/ndzip/cuda.hh:10-41: construct once with the stream, call per array with device
// pointers, nothing synchronises, failures are std::runtime_error) for the path the reference does not have: one process (or thread)
// per GPU, contiguous hypercube ranges per rank, RCCL for the offset prefix sum and the header gather only.
//
//   ndzip::hip_sharded_codec<float> codec(ndzip::extent{2048, 1024, 1024}, rank, world, nccl_comm, stream);
//   codec.compress(d_slab);                 // codec kernel + the two all-gathers, on `stream`
//   codec.check();                          // MANDATORY before the header / stream is consumed
//   codec.decompress(d_slab_out);           // no collective
//   codec.write_stream(mapped_file, codec.stream_layout().stream_words, rank == 0);
//
// Uses ndzip::extent of include/ndzip_hip.hh (or, with NDZIP_HIP_WITH_REFERENCE_HEADERS, the reference's own).
#pragma once

#include <utility>

#include "ndzip_hip.hh"
#include "ndzip_hip_sharded.h"

namespace ndzip {

namespace hip_detail {
inline void check_sharded(int status) {
    if (status != NDZIP_HIP_OK) throw std::runtime_error(std::string("ndzip_hip_sharded: ") + ndzip_hip_sharded_last_error());
}
}  // namespace hip_detail

template<typename T>
class hip_sharded_codec {
  public:
    using value_type = T;
    using compressed_type = ndzip::compressed_type<T>;

    // `nccl_comm`: the caller's ncclComm_t for the `world` ranks (may be null for one shard); `hip_stream`: a hipStream_t
    hip_sharded_codec(const extent &global_size, index_type rank, index_type world, void *nccl_comm, void *hip_stream = nullptr)
        : _dims(global_size.dimensions()) {
        hip_detail::check_sharded(ndzip_hip_sharded_create(
                hip_detail::dtype_of<T>(), global_size.dimensions(), global_size.begin(), rank, world, nccl_comm, hip_stream, &_handle));
        hip_detail::check_sharded(ndzip_hip_sharded_shard(_handle, &_shard));
    }
    // the same over a caller-supplied exchange (MPI_Allgather on device pointers, ...)
    hip_sharded_codec(const extent &global_size, index_type rank, index_type world, const ndzip_hip_collectives &exchange, void *hip_stream = nullptr)
        : _dims(global_size.dimensions()) {
        hip_detail::check_sharded(ndzip_hip_sharded_create_with_collectives(
                hip_detail::dtype_of<T>(), global_size.dimensions(), global_size.begin(), rank, world, &exchange, hip_stream, &_handle));
        hip_detail::check_sharded(ndzip_hip_sharded_shard(_handle, &_shard));
    }
    hip_sharded_codec(const hip_sharded_codec &) = delete;
    hip_sharded_codec &operator=(const hip_sharded_codec &) = delete;
    ~hip_sharded_codec() { ndzip_hip_sharded_destroy(_handle); }

    // this rank's part of the plan: rows [first_row(), first_row() + local_size()[0]) of dimension 0, the other dimensions whole
    const ndzip_hip_shard &shard() const { return _shard; }
    index_type first_row() const { return _shard.start0; }
    extent local_size() const {
        extent e(_dims);
        for (dim_type d = 0; d < _dims; ++d) e[d] = _shard.extent[d];
        return e;
    }

    void compress(const value_type *in_device_slab) { hip_detail::check_sharded(ndzip_hip_sharded_compress(_handle, in_device_slab)); }
    // the two halves of compress, for a host that wants something between them (events, a decode from the local offsets)
    void compress_local(const value_type *in_device_slab) { hip_detail::check_sharded(ndzip_hip_sharded_compress_local(_handle, in_device_slab)); }
    void exchange() { hip_detail::check_sharded(ndzip_hip_sharded_exchange(_handle)); }
    void decompress(value_type *out_device_slab) { hip_detail::check_sharded(ndzip_hip_sharded_decompress(_handle, out_device_slab)); }
    // sticky device error words (look-back time-out, corrupt header, offsets beyond 32 bits); synchronises the stream
    void check() { hip_detail::check_sharded(ndzip_hip_sharded_check(_handle)); }

    // device-resident results of the last compress
    const index_type *header_global(index_type *num_entries = nullptr) const {
        const uint32_t *p = nullptr;
        hip_detail::check_sharded(ndzip_hip_sharded_header_global(_handle, &p, num_entries));
        return p;
    }
    const compressed_type *body(const index_type **device_length_words = nullptr, const index_type **device_base_words = nullptr) const {
        const void *p = nullptr;
        hip_detail::check_sharded(ndzip_hip_sharded_body(_handle, &p, device_length_words, device_base_words));
        return static_cast<const compressed_type *>(p);
    }

    // the reference's single stream of the global array: where this rank's pieces go, and the copies (file level, synchronising)
    ndzip_hip_stream_layout stream_layout() {
        ndzip_hip_stream_layout l{};
        hip_detail::check_sharded(ndzip_hip_sharded_stream_layout(_handle, &l));
        return l;
    }
    void write_stream(compressed_type *host_stream, uint64_t capacity_words, bool with_header) {
        hip_detail::check_sharded(ndzip_hip_sharded_write_stream(_handle, host_stream, capacity_words, with_header ? 1 : 0));
    }
    void load(const compressed_type *host_stream, uint64_t stream_words) { hip_detail::check_sharded(ndzip_hip_sharded_load(_handle, host_stream, stream_words)); }

    ndzip_hip_sharded *native_handle() { return _handle; }

  private:
    dim_type _dims;
    ndzip_hip_sharded *_handle = nullptr;
    ndzip_hip_shard _shard{};
};

// BASELINE.json's <T, Dims> spelling
template<typename T, dim_type Dims>
class hip_sharded_codec_nd : public hip_sharded_codec<T> {
  public:
    template<typename... Args>
    explicit hip_sharded_codec_nd(const extent &global_size, Args &&...args) : hip_sharded_codec<T>(check_dims(global_size), std::forward<Args>(args)...) {}

  private:
    static const extent &check_dims(const extent &e) {
        if (e.dimensions() != Dims) throw std::runtime_error("data dimensionality does not match compressor dimensionality");
        return e;
    }
};

}  // namespace ndzip
