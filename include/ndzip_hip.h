/*
 * include/ndzip_hip.h -- C ABI of the MI355X (gfx950) back-end for ndzip's block encode/decode path.
 *
 * This is the drop-in boundary: plain pointers and sizes, no C++ or torch types.  Every entry point names
 * the reference interface it replaces (file:line relative to the celerity/ndzip tree).  The C++ adaptor
 * classes with the reference's own virtual interfaces live in include/ndzip_hip.hh; the binding a
 * maintainer would add to the reference is shown in INTEGRATION.md.
 *
 * Conventions
 *   - dtype: NDZIP_HIP_F32 (stream words are uint32_t) or NDZIP_HIP_F64 (stream words are uint64_t).
 *   - dims in {1,2,3}; extent[d], d = 0 slowest (include/ndzip/ndzip.hh:172-180); all counts are uint32_t as
 *     in the format (ndzip.hh:20).  Lengths are in stream words of the dtype.
 *   - Every function returns NDZIP_HIP_OK (0) or a negative ndzip_hip_status; no exception crosses this ABI.
 *     ndzip_hip_last_error() returns a thread-local description of the last failure.
 *   - Handles are not thread-safe (same contract as the reference objects, which own mutable scratch:
 *     src/ndzip/cuda_codec.inl:536-539); distinct handles are independent.
 *   - Device-pointer entry points only enqueue work on the handle's hipStream_t; they never synchronise.
 */
#ifndef NDZIP_HIP_H
#define NDZIP_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(NDZIP_HIP_BUILD)
#define NDZIP_HIP_API __attribute__((visibility("default")))
#else
#define NDZIP_HIP_API
#endif

typedef enum ndzip_hip_dtype { NDZIP_HIP_F32 = 0, NDZIP_HIP_F64 = 1 } ndzip_hip_dtype;

typedef enum ndzip_hip_status {
    NDZIP_HIP_OK = 0,
    NDZIP_HIP_ERR_INVALID_ARGUMENT = -1, /* bad dtype / dims / null pointer */
    NDZIP_HIP_ERR_DIMS_MISMATCH = -2,    /* "data dimensionality does not match compressor dimensionality"
                                            (std::runtime_error at src/ndzip/cuda_codec.inl:557-559, :631-633) */
    NDZIP_HIP_ERR_CAPACITY = -3,         /* extent needs more hypercubes than the handle was created for */
    NDZIP_HIP_ERR_RUNTIME = -4,          /* HIP runtime failure (cuda_check, src/ndzip/cuda_bits.cuh:165-169) */
    NDZIP_HIP_ERR_NO_DEVICE = -5,        /* no gfx950 device visible: this back-end has NO CPU fallback */
    NDZIP_HIP_ERR_DEVICE_FAULT = -6,     /* sticky device-side error word set (scan timeout / corrupt header) */
    NDZIP_HIP_ERR_LIMIT = -7             /* element or word count does not fit the format's uint32_t */
} ndzip_hip_status;

typedef struct ndzip_hip_compressor ndzip_hip_compressor;
typedef struct ndzip_hip_decompressor ndzip_hip_decompressor;

/* Human-readable description of the last error on the calling thread ("" if none). */
NDZIP_HIP_API const char *ndzip_hip_last_error(void);

/* Version of THIS interface the library was built from.  It changes whenever the signature or the meaning of an existing entry
 * point does (2: ndzip_hip_compressor_offset_header_gathered took `world`), not when entry points are added: a binding generated
 * from one header and loaded against a library of another must compare it with NDZIP_HIP_ABI_VERSION before the first call
 * (ndzip_amd/hip.py does; a C or C++ program that links the library gets the check from the linker and the header together). */
#define NDZIP_HIP_ABI_VERSION 2
NDZIP_HIP_API int ndzip_hip_abi_version(void);

/* Library / device identification: writes the gfx arch name of the current device (e.g. "gfx950") and its CU
 * count.  Fails with NDZIP_HIP_ERR_NO_DEVICE when no GPU is visible. */
NDZIP_HIP_API int ndzip_hip_device_info(char *arch, size_t arch_capacity, int *num_compute_units);

/* ---- sizing -------------------------------------------------------------------------------------------- */

/* ndzip::compressed_length_bound<T>(extent)  (include/ndzip/ndzip.hh:224-225, src/ndzip/common.cc:31-55) */
NDZIP_HIP_API int ndzip_hip_compressed_length_bound(int dtype, int dims, const uint32_t *extent, uint64_t *words);

/* detail::num_hypercubes(extent)  (src/ndzip/common.hh:395-412); what compressor_requirements accumulates
 * (include/ndzip/ndzip.hh:255-269, src/ndzip/common.cc:8-28). */
NDZIP_HIP_API int ndzip_hip_num_hypercubes(int dims, const uint32_t *extent, uint32_t *num_hypercubes);

/* Words of stream header for `num_hypercubes` (stream<Profile>::hypercube(0) - buffer, common.hh:350-358). */
NDZIP_HIP_API int ndzip_hip_header_words(int dtype, uint32_t num_hypercubes, uint32_t *words);

/* ---- device-pointer interface -------------------------------------------------------------------------- */

/* make_cuda_compressor<T>(const compressor_requirements&, cudaStream_t)  (include/ndzip/cuda.hh:36-38,
 * src/ndzip/cuda_factory.cu:4-9).  `max_num_hypercubes` is what compressor_requirements carries; scratch is
 * allocated once here and reused by every compress call (cuda_codec.inl:543-552).  `hip_stream` is a
 * hipStream_t (NULL = default stream). */
NDZIP_HIP_API int ndzip_hip_compressor_create(
        int dtype, int dims, uint32_t max_num_hypercubes, void *hip_stream, ndzip_hip_compressor **out);

/* cuda_compressor<T>::compress(in_device_data, data_size, out_device_stream, out_device_stream_length)
 * (include/ndzip/cuda.hh:10-23, src/ndzip/cuda_codec.inl:554-603).  Asynchronous on the handle's stream.
 * `d_stream` must hold ndzip_hip_compressed_length_bound words; `d_stream_length_words` may be NULL.
 * Pure stream work: one kernel (two with a border), no allocation, no host synchronisation and no per-call state on the host
 * (what one launch hands to the next on the handle's scratch -- ticket counters, the tile descriptors' epoch -- is kept and
 * advanced on the device).  A call may therefore be recorded into a hipGraph and replayed on new data in the same buffers
 * (tests/test_hip_graph.py); the same holds for the decompress entry points, which keep no state at all.
 * Before a capture: make ONE identical un-captured call (same extent, same pointer alignment).  The first launch of each kernel
 * variant on a device (aligned / element-aligned pointers, paired / unpaired tiles) resolves its occupancy and function
 * attributes through the runtime and caches them in the library; and what a capture records -- kernel variant, grid size -- is
 * chosen from the extent and the pointers' alignment, so a replay is valid for buffers of the same extent and alignment only. */
NDZIP_HIP_API int ndzip_hip_compressor_compress(ndzip_hip_compressor *c, const void *d_in, int dims, const uint32_t *extent,
        void *d_stream, uint32_t *d_stream_length_words);

/* Same, with header and body written to separate device buffers and offsets local to this call: the
 * building block of the multi-GPU path (no reference counterpart).  `d_header` receives
 * num_hypercubes uint32 offset_after entries relative to `d_body`; `d_body_length_words` the body length
 * (hypercube bodies followed by this extent's border). */
NDZIP_HIP_API int ndzip_hip_compressor_compress_split(ndzip_hip_compressor *c, const void *d_in, int dims,
        const uint32_t *extent, uint32_t *d_header, void *d_body, uint32_t *d_body_length_words);

/* Adds `base` to `count` device-resident header entries (shard-local -> global offsets), asynchronously on
 * the handle's stream. */
NDZIP_HIP_API int ndzip_hip_compressor_offset_header(ndzip_hip_compressor *c, uint32_t *d_header, uint32_t count, uint32_t base);

/* Same with the base read from device memory (*d_base) when the kernel runs, so the offset exchange of the
 * multi-GPU path needs no host synchronisation. */
NDZIP_HIP_API int ndzip_hip_compressor_offset_header_device(
        ndzip_hip_compressor *c, uint32_t *d_header, uint32_t count, const uint32_t *d_base);

/* The same with the base computed on the device from the all-gathered shard lengths (`world` entries each): base of shard
 * `rank` = sum over r < rank of (d_lengths[r] - d_borders[r]) (words written by compress_split incl. the shard's border, minus
 * its border words); the base is also stored to *d_base_out (may be NULL) for ndzip_hip_decompressor_decompress_split.  One
 * launch between the two collectives of the multi-GPU path, no host synchronisation.
 * Stream offsets are index_type = uint32 (include/ndzip/ndzip.hh:20): the sums are taken in 64 bits over ALL `world` shards, and
 * when the hypercube runs of the whole plan exceed 2^32 - 1 words the handle's error word gets the bit that
 * ndzip_hip_compressor_check reports as "sharded stream exceeds the format's 32-bit offsets" -- on every rank alike -- and the
 * entries are left LOCAL with *d_base_out = 0 (the rank can still decode its own slab; nothing wrapped is ever published).
 * ndzip_hip_compressor_check is therefore MANDATORY before the globalised header is consumed. */
NDZIP_HIP_API int ndzip_hip_compressor_offset_header_gathered(ndzip_hip_compressor *c, uint32_t *d_header, uint32_t count,
        const uint32_t *d_lengths, const uint32_t *d_borders, uint32_t rank, uint32_t world, uint32_t *d_base_out);

/* Reads and clears the handle's sticky device error word; synchronises the handle's stream.  MANDATORY at the caller's
 * first host synchronisation after a compress call when the stream is going to be kept: a look-back timeout (a device
 * that made no forward progress for ~0.2 s) leaves a stream whose offsets are wrong.  As a second line of defence such a
 * launch also stores 0 to *d_stream_length_words -- shorter than any valid stream, so every consumer of the length
 * (ndzip_hip_stream_words, the decompress entry points) rejects it.  The host-pointer entry points below check for the
 * caller. */
NDZIP_HIP_API int ndzip_hip_compressor_check(ndzip_hip_compressor *c);

/* No reference counterpart (a tuning / diagnosis handle): caps the workgroups per compute unit of the persistent compress grid of
 * this handle's later launches.  0 (the default) = as many as are resident (4); 1..3 launch a smaller grid (values up to 64 are accepted and have no effect beyond what is resident) -- same kernels, same
 * stream, bit for bit -- e.g. to measure what the 4th workgroup per CU buys, or to leave room on a GPU shared with other work. */
NDZIP_HIP_API int ndzip_hip_compressor_set_max_workgroups_per_cu(ndzip_hip_compressor *c, int max_workgroups_per_cu);

NDZIP_HIP_API int ndzip_hip_compressor_destroy(ndzip_hip_compressor *c);

/* make_cuda_decompressor<T>(dims, cudaStream_t)  (include/ndzip/cuda.hh:40-41, src/ndzip/cuda_factory.cu:11-14) */
NDZIP_HIP_API int ndzip_hip_decompressor_create(int dtype, int dims, void *hip_stream, ndzip_hip_decompressor **out);

/* cuda_decompressor<T>::decompress(in_device_stream, out_device_data, data_size)
 * (include/ndzip/cuda.hh:25-34, src/ndzip/cuda_codec.inl:628-652).  Asynchronous on the handle's stream.
 * Encoded runs are fetched as 16-byte aligned blocks: the kernel may LOAD (never use, never store) up to 12 bytes in front
 * of the first and behind the last word of the stream's bodies, inside the aligned 16-byte block that holds that word --
 * harmless for any device allocation (such a block cannot straddle a page). */
NDZIP_HIP_API int ndzip_hip_decompressor_decompress(
        ndzip_hip_decompressor *d, const void *d_stream, void *d_out, int dims, const uint32_t *extent);

/* The same for a caller that knows how many words `d_stream` holds (the reference interface has no such argument and
 * trusts the header, cuda_codec.inl:628-652): header entries that point outside the stream make the affected hypercubes
 * decode as zeros and set the error word (ndzip_hip_decompressor_check) instead of reading out of bounds.  Without a
 * length (the call above) entries are still held to the format's own bounds: offset_after(hc) within
 * [(hc + 1) * 4096 / B, (hc + 1) * (4096 + 4096 / B)] words. */
NDZIP_HIP_API int ndzip_hip_decompressor_decompress_bounded(ndzip_hip_decompressor *d, const void *d_stream,
        uint32_t stream_length_words, void *d_out, int dims, const uint32_t *extent);

/* Split-buffer variant matching ndzip_hip_compressor_compress_split: `d_header` holds this extent's
 * num_hypercubes global offset_after entries; `d_header_base` points to a DEVICE uint32 holding the global
 * offset of `d_body`'s first word (NULL = 0).  For shard r > 0 that is simply the address of the previous
 * shard's last header entry, so no host synchronisation is needed. */
NDZIP_HIP_API int ndzip_hip_decompressor_decompress_split(ndzip_hip_decompressor *d, const uint32_t *d_header,
        const uint32_t *d_header_base, const void *d_body, void *d_out, int dims, const uint32_t *extent);

/* ... with `body_words` = words `d_body` holds (this shard's hypercube runs + its border) */
NDZIP_HIP_API int ndzip_hip_decompressor_decompress_split_bounded(ndzip_hip_decompressor *d, const uint32_t *d_header,
        const uint32_t *d_header_base, const void *d_body, uint32_t body_words, void *d_out, int dims, const uint32_t *extent);

/* Tuning / A-B switch, no reference counterpart (the reference fixes 512 threads per 64-bit hypercube,
 * src/ndzip/gpu_common.hh:38-43): work-items the kernel of a 64-bit profile decodes one hypercube with -- 0 = the library's
 * default (128 until the 256-work-item kernel has been measured on an MI355X), 128 (decompress_kernel, the mapping of the 32-bit
 * profiles, 2 wavefronts per SIMD) or 256 (decompress_kernel_wide, 5-7 wavefronts per SIMD).  Both produce the same bits.  No effect
 * on 32-bit profiles. */
NDZIP_HIP_API int ndzip_hip_decompressor_set_f64_work_items(ndzip_hip_decompressor *d, int work_items_per_hypercube);

/* Reads and clears the handle's sticky device error word (corrupt header entries); synchronises the handle's stream. */
NDZIP_HIP_API int ndzip_hip_decompressor_check(ndzip_hip_decompressor *d);
NDZIP_HIP_API int ndzip_hip_decompressor_destroy(ndzip_hip_decompressor *d);

/* ---- host-pointer interface ------------------------------------------------------------------------------ */

/* offloader<T>::compress(data, data_size, stream, kernel_duration*)  (include/ndzip/offload.hh:16-19;
 * behaviour of cuda_offloader::do_compress, src/ndzip/cuda_codec.inl:669-714): H2D copy, device pipeline timed
 * with events (kernel_ns, may be NULL), D2H copy of length and stream.  `stream` must hold
 * compressed_length_bound words; *stream_length_words receives the return value of the reference call.
 * A launch whose device-wide scan timed out (NDZIP_HIP_ERR_DEVICE_FAULT, "scan look-back timeout": the persistent grid was not
 * fully resident, e.g. on a GPU shared with another process) is repeated ONCE -- the array is still on the device; the retry runs
 * one workgroup per CU, a grid a shared GPU is far more likely to hold in full -- before the error is returned; kernel_ns is that of the last launch. */
NDZIP_HIP_API int ndzip_hip_offload_compress(int dtype, int dims, const uint32_t *extent, const void *data, void *stream,
        uint32_t *stream_length_words, uint64_t *kernel_ns);

/* offloader<T>::decompress(stream, length, data, data_size, kernel_duration*)  (offload.hh:21-24;
 * cuda_offloader::do_decompress, cuda_codec.inl:716-761).  *words_consumed receives the reference's return
 * value (border offset + border words). */
NDZIP_HIP_API int ndzip_hip_offload_decompress(int dtype, int dims, const uint32_t *extent, const void *stream,
        uint32_t stream_length_words, void *data, uint32_t *words_consumed, uint64_t *kernel_ns);

/* ---- persistent, pipelined host-pointer interface -------------------------------------------------------------
 * What an `offloader<T>` OBJECT is in the reference (include/ndzip/offload.hh:8-34: created once by make_offloader,
 * called per array) for callers that stream many arrays through the device -- the CLI's chunk loop
 * (src/compress/compress.cc:17-86) and the interconnect use case of README.md:13-14.  The handle owns `slots` job slots;
 * each slot has its own HIP stream, device input / stream / length buffers sized for `max_extent`, and codec handles, so
 * the H2D copy of job j+1, the kernels of job j and the D2H copy of job j-1 overlap.  Host buffers should come from
 * ndzip_hip_host_alloc (pinned: asynchronous DMA); pageable pointers work but the copies then serialise.
 * A slot is busy from submit until wait; one thread drives a handle. */
typedef struct ndzip_hip_offloader ndzip_hip_offloader;

NDZIP_HIP_API int ndzip_hip_offloader_create(int dtype, int dims, const uint32_t *max_extent, int slots, ndzip_hip_offloader **out);
NDZIP_HIP_API int ndzip_hip_offloader_destroy(ndzip_hip_offloader *o);

/* pinned host memory (hipHostMalloc / hipHostFree) */
NDZIP_HIP_API int ndzip_hip_host_alloc(size_t bytes, void **ptr);
NDZIP_HIP_API int ndzip_hip_host_free(void *ptr);

/* enqueue offloader<T>::compress(data, extent, stream) on `slot`: returns at once; `data` and `stream`
 * (compressed_length_bound words) must stay valid until the wait */
NDZIP_HIP_API int ndzip_hip_offloader_submit_compress(ndzip_hip_offloader *o, int slot, const uint32_t *extent, const void *data,
        void *stream);
/* enqueue offloader<T>::decompress(stream, length, data, extent) on `slot` */
NDZIP_HIP_API int ndzip_hip_offloader_submit_decompress(ndzip_hip_offloader *o, int slot, const uint32_t *extent,
        const void *stream, uint32_t stream_length_words, void *data);
/* complete the job of `slot`: *words = the reference call's return value (stream length / words consumed), *kernel_ns =
 * device pipeline time by events (either may be NULL).  For a compress job the stream's D2H copy happens here, once the
 * length is known, on the slot's own stream; a compress job whose scan timed out is relaunched once from the slot's device copy of
 * the array (see ndzip_hip_offload_compress) before NDZIP_HIP_ERR_DEVICE_FAULT is returned. */
NDZIP_HIP_API int ndzip_hip_offloader_wait(ndzip_hip_offloader *o, int slot, uint32_t *words, uint64_t *kernel_ns);

/* Words of the stream of an array of `extent` that starts at HOST pointer `stream`: header + last offset + border
 * (what decompress returns, cuda_codec.inl:740-745), from the header alone -- lets a reader split a file of concatenated
 * streams (src/compress/compress.cc:62-86) before decompressing.  `available_words`: how many words `stream` holds.
 * Validates the whole header: every entry must follow its predecessor by the length of one encoded hypercube (4096 / B
 * .. 4096 + 4096 / B words) and the implied stream must fit `available_words`; NDZIP_HIP_ERR_INVALID_ARGUMENT otherwise.
 * The host-pointer decompress entry points call this first, so a corrupt or truncated stream never reaches the device. */
NDZIP_HIP_API int ndzip_hip_stream_words(int dtype, int dims, const uint32_t *extent, const void *stream, uint64_t available_words,
        uint32_t *words);

/* ---- arrays beyond the format's 32-bit counts ------------------------------------------------------------------------------
 * index_type is uint32_t (include/ndzip/ndzip.hh:20): one ndzip stream holds fewer than 2^32 elements and fewer than 2^32
 * words.  The reference handles larger data at the file level only -- its tool cuts the input into arrays of `-n` elements and
 * concatenates one stream per array (src/compress/compress.cc:34-45).  These entry points do the same for ONE large host
 * array: dimension 0 (given in 64 bits) is cut into the fewest slabs of whole hypercube rows that are legal ndzip arrays,
 * slab k covers rows [k * rows_per_chunk, min((k + 1) * rows_per_chunk, extent[0])), every slab becomes an independent
 * stream -- bit-identical to what the reference produces for that slab as an array of its own -- and the streams are
 * concatenated in slab order.  Two slabs are in flight on the device (ndzip_hip_offloader_*).
 * `max_elements`: 0 = the format's limit; a smaller value forces more slabs (tests, or bounding device memory). */

/* the plan for `extent` (extent[0] may exceed 32 bits): rows of dimension 0 per slab, number of slabs, and the words the
 * concatenated streams can take at most (any of the three outputs may be NULL) */
NDZIP_HIP_API int ndzip_hip_chunked_plan(int dtype, int dims, const uint64_t *extent, uint64_t max_elements, uint64_t *rows_per_chunk,
        uint64_t *num_chunks, uint64_t *length_bound_words);

/* `streams` has room for `capacity_words`: the plan's length bound always suffices, and any capacity that holds the streams
 * actually produced is enough (each slab is copied from the device straight behind its predecessors once its exact length is
 * known; NDZIP_HIP_ERR_CAPACITY otherwise).  *total_words = words of the concatenation */
NDZIP_HIP_API int ndzip_hip_chunked_compress(int dtype, int dims, const uint64_t *extent, uint64_t max_elements, const void *data,
        void *streams, uint64_t capacity_words, uint64_t *total_words, uint64_t *kernel_ns);

/* every slab's header is validated (ndzip_hip_stream_words) before it is sent to the device */
NDZIP_HIP_API int ndzip_hip_chunked_decompress(int dtype, int dims, const uint64_t *extent, uint64_t max_elements, const void *streams,
        uint64_t total_words, void *data, uint64_t *words_consumed, uint64_t *kernel_ns);

/* The stage entry points of the parity tests (one hypercube through the kernels' device functions) are not part of this
 * interface: include/ndzip_hip_stages.h, libndzip_hip_stages.so. */

#ifdef __cplusplus
}
#endif

#endif /* NDZIP_HIP_H */
