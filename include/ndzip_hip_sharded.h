/*
 * include/ndzip_hip_sharded.h -- C ABI of the multi-GPU path: one process (or thread) per GPU, contiguous hypercube ranges
 * per rank, RCCL over xGMI for the offset prefix sum and the header gather ONLY (libndzip_hip_rccl.so).
 *
 * The reference has no distributed runtime; its device-pointer interface -- cuda_compressor<T>::compress /
 * cuda_decompressor<T>::decompress on a caller's stream (include/ndzip/cuda.hh:10-41), meant for compressing data on its way
 * over an interconnect (README.md:13-14) -- is the interface style kept here: device pointers in, device-resident results
 * out, work enqueued on the caller's hipStream_t, no host synchronisation on the data path.
 *
 * The path:
 *   plan      dimension 0 (slowest) of the global array is cut into `world` slabs of whole hypercube planes (the last slab also
 *             takes the rows no hypercube covers); hypercube order is row-major over the hypercube grid
 *             (src/ndzip/common.hh:414-433), so slab r owns the contiguous hypercube range [hc_begin, hc_end) and needs no halo.
 *   compress  (1) ndzip_hip_compressor_compress_split of the slab: header entries LOCAL to this rank's body;
 *             (2) all-gather of ONE uint32 per rank: words written (hypercube runs + the slab's border);
 *             (3) one kernel: base_r = sum_{q<r} (words_q - border_q) taken in 64 bits over ALL ranks, added to the rank's header
 *                 entries; a plan whose hypercube runs exceed the format's uint32 offsets (include/ndzip/ndzip.hh:20) sets the
 *                 error word on EVERY rank (ndzip_hip_sharded_check);
 *             (4) all-gather of the header segments.  Bodies never move.
 *   stream    the reference's single stream is [header][body_0]...[body_{R-1}][border_0]...[border_{R-1}];
 *             ndzip_hip_sharded_stream_layout tells a rank where its three pieces go, ndzip_hip_sharded_write_stream copies them
 *             there (a file-level step, outside any timed region).
 *   decompress  no collective: a rank decodes its slab from its own header entries, its base and its resident body.
 *
 * libndzip_hip_rccl.so links libndzip_hip.so (include/ndzip_hip.h: the kernels) and librccl; the product library itself stays
 * free of RCCL.  The exchange sits behind a one-function table so that a host with another transport (MPI, or the gloo-backed
 * table of this repository's CPU tests) can drive the same path.  Return values and the handle rules are those of
 * include/ndzip_hip.h; error strings: ndzip_hip_sharded_last_error.
 */
#ifndef NDZIP_HIP_SHARDED_H
#define NDZIP_HIP_SHARDED_H

#include "ndzip_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

#define NDZIP_HIP_SHARDED_ABI_VERSION 1

typedef struct ndzip_hip_sharded ndzip_hip_sharded;

/* One rank's part of the plan. */
typedef struct ndzip_hip_shard {
    uint32_t rank, world;
    uint32_t start0;          /* first row of dimension 0 */
    uint32_t extent[3];       /* the slab: dimension 0 cut, the others whole (unused entries 0) */
    uint32_t hc_begin, hc_end; /* global hypercube index range */
    uint32_t border_elements; /* elements of the slab no hypercube covers == words of its border */
    uint64_t body_capacity_words; /* words the slab's runs + border can take at most */
} ndzip_hip_shard;

/* The exchange.  all_gather_u32: rank r's `count` uint32 at d_send land at d_recv[r * count ...] on every rank, enqueued on
 * `hip_stream` (stream-ordered like ncclAllGather: the buffers are valid when the work before it on that stream has run, and
 * the result is visible to the work after it).  Returns 0 or a transport error code, which is reported with error_string
 * (may be NULL).  `ctx` is handed back unchanged. */
typedef struct ndzip_hip_collectives {
    void *ctx;
    int (*all_gather_u32)(void *ctx, const uint32_t *d_send, uint32_t *d_recv, size_t count, void *hip_stream);
    const char *(*error_string)(void *ctx, int code);
} ndzip_hip_collectives;

NDZIP_HIP_API int ndzip_hip_sharded_abi_version(void);

/* Description of the last failure of an entry point of THIS header on the calling thread (a failure inside libndzip_hip.so
 * is passed through: the text is ndzip_hip_last_error()'s). */
NDZIP_HIP_API const char *ndzip_hip_sharded_last_error(void);

/* The plan alone (host arithmetic; no device needed): shard `rank` of `world` for `global_extent`. */
NDZIP_HIP_API int ndzip_hip_sharded_plan(int dtype, int dims, const uint32_t *global_extent, uint32_t rank, uint32_t world, ndzip_hip_shard *out);

/* Per-rank driver over RCCL.  `nccl_comm` is the caller's ncclComm_t for the `world` ranks (rccl.h); `hip_stream` the
 * hipStream_t everything is enqueued on (NULL = default stream); the current device must be the communicator's.
 * Refuses (NDZIP_HIP_ERR_LIMIT) a global extent the stream format cannot carry: more than 2^32 - 1 elements, hypercube runs
 * that could exceed the uint32 offsets, or a stream bound beyond the uint32 length word.  All buffers are allocated here. */
NDZIP_HIP_API int ndzip_hip_sharded_create(int dtype, int dims, const uint32_t *global_extent, uint32_t rank, uint32_t world, void *nccl_comm,
        void *hip_stream, ndzip_hip_sharded **out);

/* The same over a caller-supplied exchange (the table is copied). */
NDZIP_HIP_API int ndzip_hip_sharded_create_with_collectives(int dtype, int dims, const uint32_t *global_extent, uint32_t rank, uint32_t world,
        const ndzip_hip_collectives *collectives, void *hip_stream, ndzip_hip_sharded **out);

/* ---- in-process rank group: every rank a THREAD of one process ---------------------------------------------------------------
 * For a single process that drives several GPUs (one thread per GPU) -- or several ranks on one GPU -- without RCCL: the all-gather
 * is a rendezvous of the group's threads followed by device-to-device copies on each rank's own stream (peer copies over xGMI
 * between GPUs).  Rank r's thread selects its device, creates its handle with ndzip_hip_sharded_create_local and then calls the
 * same entry points as any other rank; all ranks must reach each ndzip_hip_sharded_compress / _exchange (a collective). */
typedef struct ndzip_hip_local_group ndzip_hip_local_group;
NDZIP_HIP_API int ndzip_hip_local_group_create(uint32_t world, ndzip_hip_local_group **out);
NDZIP_HIP_API int ndzip_hip_local_group_destroy(ndzip_hip_local_group *group);
/* a barrier of the group's `world` threads (for the host's own hand-offs: "every rank has written its pieces") */
NDZIP_HIP_API int ndzip_hip_local_group_barrier(ndzip_hip_local_group *group);
NDZIP_HIP_API int ndzip_hip_sharded_create_local(int dtype, int dims, const uint32_t *global_extent, uint32_t rank, uint32_t world,
        ndzip_hip_local_group *group, void *hip_stream, ndzip_hip_sharded **out);

/* hipGetDeviceCount / hipSetDevice for a host that includes no HIP header (the calling thread's current device) */
NDZIP_HIP_API int ndzip_hip_sharded_device_count(int *count);
NDZIP_HIP_API int ndzip_hip_sharded_set_device(int device);

/* Bootstrap helpers for a host that does not want rccl.h itself: ncclGetUniqueId (rank 0; send the 128 bytes to the other
 * ranks by any means), ncclCommInitRank (every rank, after hipSetDevice), ncclCommDestroy. */
#define NDZIP_HIP_RCCL_UNIQUE_ID_BYTES 128
NDZIP_HIP_API int ndzip_hip_rccl_unique_id(void *id_bytes);
NDZIP_HIP_API int ndzip_hip_rccl_comm_create(const void *id_bytes, uint32_t rank, uint32_t world, void **nccl_comm);
NDZIP_HIP_API int ndzip_hip_rccl_comm_destroy(void *nccl_comm);

NDZIP_HIP_API int ndzip_hip_sharded_shard(const ndzip_hip_sharded *s, ndzip_hip_shard *out);

/* d_in_slab: this rank's slab (shard.extent, row-major, device memory).  Enqueues steps (1)-(4); afterwards (stream-wise) the
 * handle holds the global header, this rank's body and base.  No host synchronisation. */
NDZIP_HIP_API int ndzip_hip_sharded_compress(ndzip_hip_sharded *s, const void *d_in_slab);

/* The two halves of it, for a host that wants something between them (events around the codec launch, a decode of the slab
 * from its local offsets in front of the exchange): step (1), then steps (2)-(4).  Exactly one exchange per compress_local;
 * until it has run, the global header is not there yet, stream_layout / write_stream are refused and
 * ndzip_hip_sharded_decompress decodes from the local offsets. */
NDZIP_HIP_API int ndzip_hip_sharded_compress_local(ndzip_hip_sharded *s, const void *d_in_slab);
NDZIP_HIP_API int ndzip_hip_sharded_exchange(ndzip_hip_sharded *s);

/* Decodes this rank's slab from what the last compress left (or ndzip_hip_sharded_load put) in the handle.  No collective. */
NDZIP_HIP_API int ndzip_hip_sharded_decompress(ndzip_hip_sharded *s, void *d_out_slab);

/* Host-pointer forms (what offloader<T> is to cuda_compressor<T>, include/ndzip/offload.hh:8-34): the slab is copied to / from a
 * device buffer the handle owns (allocated on first use).  compress_host copies in with hipMemcpyAsync (which returns early only for pinned memory: the
 * runtime's rule) and enqueues codec launch and exchange behind it; decompress_host returns when the slab is in `host_slab`. */
NDZIP_HIP_API int ndzip_hip_sharded_compress_host(ndzip_hip_sharded *s, const void *host_slab);
/* copy-in + ndzip_hip_sharded_compress_local: ndzip_hip_sharded_exchange is the caller's next step */
NDZIP_HIP_API int ndzip_hip_sharded_compress_local_host(ndzip_hip_sharded *s, const void *host_slab);
NDZIP_HIP_API int ndzip_hip_sharded_decompress_host(ndzip_hip_sharded *s, void *host_slab);

/* Device-resident results of the last compress: all num_hypercubes(global extent) header entries with global offsets ... */
NDZIP_HIP_API int ndzip_hip_sharded_header_global(const ndzip_hip_sharded *s, const uint32_t **d_header, uint32_t *num_entries);
/* ... and this rank's body (hypercube runs, then the slab's border), the device word holding its length in words, and the
 * device word holding the global word offset of its first run (relative to the end of the header).  Any may be NULL. */
NDZIP_HIP_API int ndzip_hip_sharded_body(const ndzip_hip_sharded *s, const void **d_body, const uint32_t **d_body_length_words,
        const uint32_t **d_base_words);

/* Where this rank's pieces lie in the single stream of the global array, in words of the dtype from the stream's first word:
 * the whole header (every rank holds it), this rank's hypercube runs, this rank's border.  Synchronises the stream and reads
 * the gathered lengths; checks the error word first (a stream with wrapped offsets has no layout). */
typedef struct ndzip_hip_stream_layout {
    uint64_t header_words;
    uint64_t runs_offset_words, runs_words;
    uint64_t border_offset_words, border_words;
    uint64_t stream_words; /* of the whole stream: what the reference's compress() returns for the global array */
} ndzip_hip_stream_layout;
NDZIP_HIP_API int ndzip_hip_sharded_stream_layout(ndzip_hip_sharded *s, ndzip_hip_stream_layout *out);

/* Copies this rank's pieces to their place in `host_stream` (a host buffer, or a shared mapping of the output file, of
 * layout.stream_words words): runs and border always, the header when `with_header` is non-zero (one rank does). */
NDZIP_HIP_API int ndzip_hip_sharded_write_stream(ndzip_hip_sharded *s, void *host_stream, uint64_t capacity_words, int with_header);

/* The way back: takes this rank's pieces out of a single stream of the global array in host memory (validated with
 * ndzip_hip_stream_words first) and uploads them, so that ndzip_hip_sharded_decompress decodes the slab.  No collective. */
NDZIP_HIP_API int ndzip_hip_sharded_load(ndzip_hip_sharded *s, const void *host_stream, uint64_t stream_words);

/* Sticky device error words of the rank's codec handles (look-back time-out, corrupt header, "sharded stream exceeds the
 * format's 32-bit offsets"); synchronises the stream.  MANDATORY before the global header or the stream is consumed. */
NDZIP_HIP_API int ndzip_hip_sharded_check(ndzip_hip_sharded *s);

NDZIP_HIP_API int ndzip_hip_sharded_destroy(ndzip_hip_sharded *s);

#ifdef __cplusplus
}
#endif

#endif /* NDZIP_HIP_SHARDED_H */
