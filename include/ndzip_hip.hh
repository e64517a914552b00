// include/ndzip_hip.hh -- header-only C++ adaptor over the C ABI (ndzip_hip.h) with the reference's own interfaces.
//
// A user of celerity/ndzip switches back-end by swapping a factory call:
//
//   reference (include/ndzip/...)                                     this header (namespace ndzip)
//   ------------------------------------------------------------------------------------------------------------
//   make_cuda_compressor<T>(const compressor_requirements&, stream)   make_hip_compressor<T>(req, hipStream_t)
//   cuda_compressor<T>::compress(in, extent, out, out_len)  cuda.hh   hip_compressor<T>::compress(...)   same args
//   make_cuda_decompressor<T>(dims, stream)                           make_hip_decompressor<T>(dims, hipStream_t)
//   cuda_decompressor<T>::decompress(in, out, extent)       cuda.hh   hip_decompressor<T>::decompress(...)
//   make_offloader<T>(target::cuda, dims)                 offload.hh  make_hip_offloader<T>(dims)
//   offloader<T>::compress / decompress                   offload.hh  hip_offloader<T>::compress / decompress
//   compressed_length_bound<T>(extent)                     ndzip.hh   hip_compressed_length_bound<T>(extent)
//   make_compressor<T>(dims, threads) / make_decompressor<T>  ndzip.hh make_hip_host_compressor<T>(dims) / _decompressor<T>
//   compressor<T>::compress(data, extent, stream)           ndzip.hh  hip_host_compressor<T>   : compressor<T>
//   decompressor<T>::decompress(stream, data, extent)       ndzip.hh  hip_host_decompressor<T> : decompressor<T>
//   compressor<T, Dims> / decompressor<T, Dims>  (BASELINE north-star spelling; this reference revision only has
//   the <T> + runtime-dims form)             hip_host_compressor_nd<T, Dims>, hip_host_decompressor_nd<T, Dims>
//                                                                      (device pointers: hip_compressor_nd / hip_decompressor_nd)
//
// When built inside the reference tree, define NDZIP_HIP_WITH_REFERENCE_HEADERS before including this file: the
// adaptor then uses ndzip::extent / compressor_requirements / offloader<T> from <ndzip/ndzip.hh>, <ndzip/offload.hh>
// hip_offloader<T> derives from ndzip::offloader<T>, hip_host_compressor<T> from ndzip::compressor<T>.  Stand-alone (the default) it ships minimal equivalents
// with the same members.  Errors of the C ABI are rethrown as std::runtime_error (the reference throws
// std::runtime_error for dimensionality mismatches and device failures: cuda_codec.inl:557-559, cuda_bits.cuh:165-169).
#pragma once

#include <cassert>
#include <chrono>
#include <cstdint>
#include <initializer_list>
#include <memory>
#include <stdexcept>
#include <string>
#include <type_traits>

#include "ndzip_hip.h"

#ifdef NDZIP_HIP_WITH_REFERENCE_HEADERS
#include <ndzip/ndzip.hh>
#include <ndzip/offload.hh>
#else
namespace ndzip {

using dim_type = int;
using index_type = uint32_t;
inline constexpr dim_type max_dimensionality = 3;

// ndzip::extent (include/ndzip/ndzip.hh:35-160), the members the codec interfaces use
class extent {
  public:
    constexpr extent() noexcept = default;
    constexpr explicit extent(dim_type dims) noexcept : _dims{dims} {}
    extent(std::initializer_list<index_type> components) : _dims{static_cast<dim_type>(components.size())} {
        if (_dims < 1 || _dims > max_dimensionality) throw std::runtime_error("Invalid dimensionality");
        dim_type d = 0;
        for (auto c : components) _components[d++] = c;
    }
    static extent broadcast(dim_type dims, index_type scalar) {
        extent e(dims);
        for (dim_type d = 0; d < dims; ++d) e[d] = scalar;
        return e;
    }
    constexpr dim_type dimensions() const { return _dims; }
    index_type &operator[](dim_type d) { return _components[d]; }
    index_type operator[](dim_type d) const { return _components[d]; }
    const index_type *begin() const { return _components; }
    const index_type *end() const { return _components + _dims; }

  private:
    dim_type _dims = 1;
    index_type _components[max_dimensionality] = {};
};

template<typename Extent>
index_type num_elements(const Extent &size) {
    index_type n = 1;
    for (dim_type d = 0; d < size.dimensions(); ++d) n *= size[d];
    return n;
}

template<typename T>
using compressed_type = std::conditional_t<sizeof(T) == 4, uint32_t, uint64_t>;

using kernel_duration = std::chrono::duration<uint64_t, std::nano>;

// ndzip::compressor_requirements (ndzip.hh:255-269, common.cc:8-28)
class compressor_requirements {
  public:
    compressor_requirements() = default;
    compressor_requirements(const extent &single_data_size) { include(single_data_size); }  // NOLINT
    compressor_requirements(std::initializer_list<extent> data_sizes) {
        for (const auto &e : data_sizes) include(e);
    }
    void include(const extent &data_size) {
        if (_dims == -1) {
            _dims = data_size.dimensions();
        } else if (data_size.dimensions() != _dims) {
            throw std::runtime_error("Cannot add a " + std::to_string(data_size.dimensions()) + "-dimensional extent to "
                    + std::to_string(_dims) + "-dimensional compressor_requirements");
        }
        uint32_t nhc = 0;
        if (ndzip_hip_num_hypercubes(data_size.dimensions(), data_size.begin(), &nhc) != NDZIP_HIP_OK) {
            throw std::runtime_error(ndzip_hip_last_error());
        }
        if (nhc > _max_num_hypercubes) _max_num_hypercubes = nhc;
    }
    dim_type dimensions() const { return _dims; }
    index_type max_num_hypercubes() const { return _max_num_hypercubes; }

  private:
    dim_type _dims = -1;
    index_type _max_num_hypercubes = 0;
};

// ndzip::compressor<T> / decompressor<T> (include/ndzip/ndzip.hh:227-247): the host-pointer plugin interface every CPU
// back-end of the reference implements
template<typename T>
class compressor {
  public:
    using value_type = T;
    using compressed_type = ndzip::compressed_type<T>;
    virtual ~compressor() = default;
    virtual index_type compress(const value_type *data, const extent &data_size, compressed_type *stream) = 0;
};

template<typename T>
class decompressor {
  public:
    using value_type = T;
    using compressed_type = ndzip::compressed_type<T>;
    virtual ~decompressor() = default;
    virtual index_type decompress(const compressed_type *stream, value_type *data, const extent &data_size) = 0;
};

// ndzip::offloader<T> (include/ndzip/offload.hh:8-34)
template<typename T>
class offloader {
  public:
    using value_type = T;
    using compressed_type = ndzip::compressed_type<T>;
    virtual ~offloader() = default;
    index_type compress(const value_type *data, const extent &data_size, compressed_type *stream, kernel_duration *duration = nullptr) {
        return do_compress(data, data_size, stream, duration);
    }
    index_type decompress(const compressed_type *stream, index_type length, value_type *data, const extent &data_size,
            kernel_duration *duration = nullptr) {
        return do_decompress(stream, length, data, data_size, duration);
    }

  protected:
    virtual index_type do_compress(const value_type *, const extent &, compressed_type *, kernel_duration *) = 0;
    virtual index_type do_decompress(const compressed_type *, index_type, value_type *, const extent &, kernel_duration *) = 0;
};

}  // namespace ndzip
#endif  // NDZIP_HIP_WITH_REFERENCE_HEADERS

namespace ndzip {

namespace hip_detail {

template<typename T>
constexpr int dtype_of() {
    static_assert(std::is_same_v<T, float> || std::is_same_v<T, double>, "ndzip supports float and double");
    return std::is_same_v<T, float> ? NDZIP_HIP_F32 : NDZIP_HIP_F64;
}

inline void check(int status) {
    if (status != NDZIP_HIP_OK) throw std::runtime_error(std::string("ndzip_hip: ") + ndzip_hip_last_error());
}

#ifdef NDZIP_HIP_WITH_REFERENCE_HEADERS
inline dim_type req_dims(const compressor_requirements &r) { return detail::get_dimensionality(r); }
inline index_type req_nhc(const compressor_requirements &r) { return detail::get_num_hypercubes(r); }
#else
inline dim_type req_dims(const compressor_requirements &r) {
    if (r.dimensions() == -1) throw std::runtime_error("Cannot construct a compressor with empty requirements");  // common.hh:320
    return r.dimensions();
}
inline index_type req_nhc(const compressor_requirements &r) { return r.max_num_hypercubes(); }
#endif

}  // namespace hip_detail

// ndzip::compressed_length_bound<T>(extent) (ndzip.hh:224-225)
template<typename T>
index_type hip_compressed_length_bound(const extent &e) {
    uint64_t words = 0;
    hip_detail::check(ndzip_hip_compressed_length_bound(hip_detail::dtype_of<T>(), e.dimensions(), e.begin(), &words));
    if (words > 0xffffffffull) throw std::runtime_error("compressed length bound exceeds index_type");
    return static_cast<index_type>(words);
}

// Device-pointer compressor: the shape of ndzip::cuda_compressor<T> (include/ndzip/cuda.hh:10-23).  Asynchronous on the
// stream given at construction; `out_device_stream_length` may be nullptr.
template<typename T>
class hip_compressor {
  public:
    using value_type = T;
    using compressed_type = ndzip::compressed_type<T>;

    hip_compressor(const compressor_requirements &req, void *hip_stream = nullptr) : _dims(hip_detail::req_dims(req)) {
        hip_detail::check(ndzip_hip_compressor_create(hip_detail::dtype_of<T>(), _dims, hip_detail::req_nhc(req), hip_stream, &_handle));
    }
    hip_compressor(const hip_compressor &) = delete;
    hip_compressor &operator=(const hip_compressor &) = delete;
    virtual ~hip_compressor() { ndzip_hip_compressor_destroy(_handle); }

    virtual void compress(const value_type *in_device_data, const extent &data_size, compressed_type *out_device_stream,
            index_type *out_device_stream_length) {
        hip_detail::check(ndzip_hip_compressor_compress(
                _handle, in_device_data, data_size.dimensions(), data_size.begin(), out_device_stream, out_device_stream_length));
    }

    // Sticky device error word (look-back timeout); synchronises the stream.  compress() is asynchronous and cannot throw for a
    // device-side failure: call this at the first host synchronisation after a compress() whose stream is kept (a timed-out launch
    // also stores 0 to *out_device_stream_length, so a caller that only looks at the length still fails loudly).
    void check() { hip_detail::check(ndzip_hip_compressor_check(_handle)); }
    ndzip_hip_compressor *native_handle() { return _handle; }

  private:
    dim_type _dims;
    ndzip_hip_compressor *_handle = nullptr;
};

// ndzip::cuda_decompressor<T> (include/ndzip/cuda.hh:25-34)
template<typename T>
class hip_decompressor {
  public:
    using value_type = T;
    using compressed_type = ndzip::compressed_type<T>;

    explicit hip_decompressor(dim_type dims, void *hip_stream = nullptr) {
        hip_detail::check(ndzip_hip_decompressor_create(hip_detail::dtype_of<T>(), dims, hip_stream, &_handle));
    }
    hip_decompressor(const hip_decompressor &) = delete;
    hip_decompressor &operator=(const hip_decompressor &) = delete;
    virtual ~hip_decompressor() { ndzip_hip_decompressor_destroy(_handle); }

    virtual void decompress(const compressed_type *in_device_stream, value_type *out_device_data, const extent &data_size) {
        hip_detail::check(ndzip_hip_decompressor_decompress(_handle, in_device_stream, out_device_data, data_size.dimensions(), data_size.begin()));
    }

    void check() { hip_detail::check(ndzip_hip_decompressor_check(_handle)); }

  private:
    ndzip_hip_decompressor *_handle = nullptr;
};

// BASELINE.json spells the plugin API compressor<T, Dims> / decompressor<T, Dims>: compile-time-dims aliases.
template<typename T, dim_type Dims>
class hip_compressor_nd : public hip_compressor<T> {
  public:
    static_assert(Dims >= 1 && Dims <= 3);
    hip_compressor_nd(const compressor_requirements &req, void *hip_stream = nullptr) : hip_compressor<T>(req, hip_stream) {
        if (hip_detail::req_dims(req) != Dims) throw std::runtime_error("data dimensionality does not match compressor dimensionality");
    }
};

template<typename T, dim_type Dims>
class hip_decompressor_nd : public hip_decompressor<T> {
  public:
    static_assert(Dims >= 1 && Dims <= 3);
    explicit hip_decompressor_nd(void *hip_stream = nullptr) : hip_decompressor<T>(Dims, hip_stream) {}
};

// Host-pointer interface: ndzip::offloader<T> implemented on the GPU (behaviour of cuda_offloader, cuda_codec.inl:654-761)
template<typename T>
class hip_offloader final : public offloader<T> {
  public:
    using value_type = T;
    using compressed_type = ndzip::compressed_type<T>;
    explicit hip_offloader(dim_type dims) : _dims(dims) {
        if (dims < 1 || dims > 3) throw std::runtime_error("Invalid dimensionality");  // common.hh:642
    }

  protected:
    index_type do_compress(const value_type *data, const extent &data_size, compressed_type *stream, kernel_duration *duration) override {
        if (data_size.dimensions() != _dims) throw std::runtime_error("data dimensionality does not match compressor dimensionality");
        uint32_t words = 0;
        uint64_t ns = 0;
        hip_detail::check(ndzip_hip_offload_compress(
                hip_detail::dtype_of<T>(), _dims, data_size.begin(), data, stream, &words, duration ? &ns : nullptr));
        if (duration) *duration = kernel_duration{ns};
        return words;
    }
    index_type do_decompress(const compressed_type *stream, index_type length, value_type *data, const extent &data_size,
            kernel_duration *duration) override {
        if (data_size.dimensions() != _dims) throw std::runtime_error("data dimensionality does not match decompressor dimensionality");
        uint32_t consumed = 0;
        uint64_t ns = 0;
        hip_detail::check(ndzip_hip_offload_decompress(
                hip_detail::dtype_of<T>(), _dims, data_size.begin(), stream, length, data, &consumed, duration ? &ns : nullptr));
        if (duration) *duration = kernel_duration{ns};
        return consumed;
    }

  private:
    dim_type _dims;
};

// ndzip::compressor<T> / decompressor<T> implemented on the GPU: what `make_compressor<T>(dims)` / `make_decompressor<T>(dims)`
// (ndzip.hh:249-253, cpu_factory.cc:25-49) hand out, for callers written against the reference's plain host-pointer plugin
// interface.  Same contract: host buffers in, stream words out, std::runtime_error on a dimensionality mismatch
// (cpu_codec.inl:600-602).  decompress() has no length argument in this interface (ndzip.hh:245): the stream is sized from
// its own header, which ndzip_hip_stream_words validates entry by entry against `stream_capacity_words` (default: the
// format's bound for the extent, i.e. what the caller must have allocated for compress()).
template<typename T>
class hip_host_compressor final : public compressor<T> {
  public:
    using value_type = T;
    using compressed_type = ndzip::compressed_type<T>;
    explicit hip_host_compressor(dim_type dims) : _dims(dims) {
        if (dims < 1 || dims > 3) throw std::runtime_error("Invalid dimensionality");  // common.hh:642
    }
    index_type compress(const value_type *data, const extent &data_size, compressed_type *stream) override {
        if (data_size.dimensions() != _dims) throw std::runtime_error("data dimensionality does not match compressor dimensionality");
        uint32_t words = 0;
        hip_detail::check(ndzip_hip_offload_compress(hip_detail::dtype_of<T>(), _dims, data_size.begin(), data, stream, &words, nullptr));
        return words;
    }

  private:
    dim_type _dims;
};

template<typename T>
class hip_host_decompressor final : public decompressor<T> {
  public:
    using value_type = T;
    using compressed_type = ndzip::compressed_type<T>;
    explicit hip_host_decompressor(dim_type dims) : _dims(dims) {
        if (dims < 1 || dims > 3) throw std::runtime_error("Invalid dimensionality");
    }
    index_type decompress(const compressed_type *stream, value_type *data, const extent &data_size) override {
        if (data_size.dimensions() != _dims) throw std::runtime_error("data dimensionality does not match decompressor dimensionality");
        uint64_t capacity = 0;
        hip_detail::check(ndzip_hip_compressed_length_bound(hip_detail::dtype_of<T>(), _dims, data_size.begin(), &capacity));
        uint32_t length = 0, consumed = 0;
        hip_detail::check(ndzip_hip_stream_words(hip_detail::dtype_of<T>(), _dims, data_size.begin(), stream, capacity, &length));
        hip_detail::check(ndzip_hip_offload_decompress(hip_detail::dtype_of<T>(), _dims, data_size.begin(), stream, length, data, &consumed, nullptr));
        return consumed;
    }

  private:
    dim_type _dims;
};

// BASELINE.json's spelling compressor<T, Dims> / decompressor<T, Dims> for the host-pointer plugin interface
template<typename T, dim_type Dims>
class hip_host_compressor_nd final : public compressor<T> {
  public:
    static_assert(Dims >= 1 && Dims <= 3);
    index_type compress(const T *data, const extent &data_size, compressed_type<T> *stream) override { return _impl.compress(data, data_size, stream); }

  private:
    hip_host_compressor<T> _impl{Dims};
};

template<typename T, dim_type Dims>
class hip_host_decompressor_nd final : public decompressor<T> {
  public:
    static_assert(Dims >= 1 && Dims <= 3);
    index_type decompress(const compressed_type<T> *stream, T *data, const extent &data_size) override { return _impl.decompress(stream, data, data_size); }

  private:
    hip_host_decompressor<T> _impl{Dims};
};

// Persistent, pipelined host-pointer offloader (ndzip_hip_offloader_*): device buffers, streams and codec handles are created
// once for arrays up to `max_size`; `slots` jobs are in flight, so the H2D copy of one array, the kernels of the previous one
// and the D2H copy of the one before overlap.  It IS an offloader<T> (compress / decompress run one job on slot 0 and wait),
// and adds submit / wait for callers that keep several arrays in flight, e.g. the chunk loop of src/compress/compress.cc:34-45.
template<typename T>
class hip_pipelined_offloader final : public offloader<T> {
  public:
    using value_type = T;
    using compressed_type = ndzip::compressed_type<T>;

    hip_pipelined_offloader(const extent &max_size, int slots) : _dims(max_size.dimensions()), _slots(slots) {
        hip_detail::check(ndzip_hip_offloader_create(hip_detail::dtype_of<T>(), _dims, max_size.begin(), slots, &_h));
    }
    ~hip_pipelined_offloader() override { ndzip_hip_offloader_destroy(_h); }
    hip_pipelined_offloader(const hip_pipelined_offloader &) = delete;
    hip_pipelined_offloader &operator=(const hip_pipelined_offloader &) = delete;

    int slots() const { return _slots; }

    // `data` / `stream` must stay valid until wait(slot); use hip_host_buffer for pinned memory
    void submit_compress(int slot, const value_type *data, const extent &data_size, compressed_type *stream) {
        check_dims(data_size);
        hip_detail::check(ndzip_hip_offloader_submit_compress(_h, slot, data_size.begin(), data, stream));
    }
    void submit_decompress(int slot, const compressed_type *stream, index_type length, value_type *data, const extent &data_size) {
        check_dims(data_size);
        hip_detail::check(ndzip_hip_offloader_submit_decompress(_h, slot, data_size.begin(), stream, length, data));
    }
    // returns what the reference call returns: stream length in words / words consumed
    index_type wait(int slot, kernel_duration *duration = nullptr) {
        uint32_t words = 0;
        uint64_t ns = 0;
        hip_detail::check(ndzip_hip_offloader_wait(_h, slot, &words, duration ? &ns : nullptr));
        if (duration) *duration = kernel_duration{ns};
        return words;
    }

  protected:
    index_type do_compress(const value_type *data, const extent &data_size, compressed_type *stream, kernel_duration *duration) override {
        submit_compress(0, data, data_size, stream);
        return wait(0, duration);
    }
    index_type do_decompress(const compressed_type *stream, index_type length, value_type *data, const extent &data_size,
            kernel_duration *duration) override {
        submit_decompress(0, stream, length, data, data_size);
        return wait(0, duration);
    }

  private:
    void check_dims(const extent &e) const {
        if (e.dimensions() != _dims) throw std::runtime_error("data dimensionality does not match compressor dimensionality");
    }
    dim_type _dims;
    int _slots;
    ndzip_hip_offloader *_h = nullptr;
};

// pinned host memory for the buffers of in-flight jobs (ndzip_hip_host_alloc)
template<typename U>
class hip_host_buffer {
  public:
    explicit hip_host_buffer(size_t count) : _count(count) {
        void *p = nullptr;
        hip_detail::check(ndzip_hip_host_alloc(count * sizeof(U), &p));
        _p = static_cast<U *>(p);
    }
    ~hip_host_buffer() { ndzip_hip_host_free(_p); }
    hip_host_buffer(const hip_host_buffer &) = delete;
    hip_host_buffer &operator=(const hip_host_buffer &) = delete;
    U *data() { return _p; }
    const U *data() const { return _p; }
    size_t size() const { return _count; }

  private:
    U *_p = nullptr;
    size_t _count;
};

// factories, named after make_cuda_compressor / make_cuda_decompressor / make_cuda_offloader (cuda.hh:36-41, offload.hh:55-57)
template<typename T>
std::unique_ptr<hip_compressor<T>> make_hip_compressor(const compressor_requirements &req, void *hip_stream = nullptr) {
    return std::make_unique<hip_compressor<T>>(req, hip_stream);
}

template<typename T>
std::unique_ptr<hip_decompressor<T>> make_hip_decompressor(dim_type dims, void *hip_stream = nullptr) {
    return std::make_unique<hip_decompressor<T>>(dims, hip_stream);
}

template<typename T>
std::unique_ptr<offloader<T>> make_hip_offloader(dim_type dimensions) {
    return std::make_unique<hip_offloader<T>>(dimensions);
}

// drop-in for make_compressor<T>(dims, num_threads) / make_decompressor<T>(dims, num_threads) (ndzip.hh:249-253); the thread
// count of the CPU back-ends has no meaning here and is accepted for signature compatibility
template<typename T>
std::unique_ptr<compressor<T>> make_hip_host_compressor(dim_type dims, unsigned /* num_threads */ = 0) {
    return std::make_unique<hip_host_compressor<T>>(dims);
}

template<typename T>
std::unique_ptr<decompressor<T>> make_hip_host_decompressor(dim_type dims, unsigned /* num_threads */ = 0) {
    return std::make_unique<hip_host_decompressor<T>>(dims);
}

template<typename T>
std::unique_ptr<hip_pipelined_offloader<T>> make_hip_pipelined_offloader(const extent &max_size, int slots = 2) {
    return std::make_unique<hip_pipelined_offloader<T>>(max_size, slots);
}

}  // namespace ndzip
