// libndzip_hip_rccl.so, transport-independent part: the per-rank driver of the multi-GPU path (include/ndzip_hip_sharded.h).
//
// Plain host C++ over the device-pointer C ABI of libndzip_hip.so (include/ndzip_hip.h) and the HIP runtime's memory calls; no
// kernel lives here.  The reference has no counterpart; what this file keeps from it is the interface style of
// include/ndzip/cuda.hh:10-41 (device pointers, caller's stream, nothing synchronises) and the stream layout of
// src/ndzip/common.hh:350-358 (header of offset_after entries relative to the first hypercube run, runs, border).
// The RCCL table and the ncclComm_t constructor are in sharded_rccl.cc.
#include <hip/hip_runtime.h>

#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <memory>
#include <mutex>
#include <new>
#include <vector>

#include "../../include/ndzip_hip_sharded.h"

namespace {

constexpr uint64_t index_max = 0xffffffffull;  // index_type = uint32_t (include/ndzip/ndzip.hh:20)

thread_local char g_error[320];
thread_local bool g_error_is_ours = false;

}  // namespace

// (library-internal, hidden: sharded_rccl.cc reports through it too, so that ndzip_hip_sharded_last_error covers both files)
int ndzip_sharded_fail(int status, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_error, sizeof g_error, fmt, ap);
    va_end(ap);
    g_error_is_ours = true;
    return status;
}

namespace {

#define fail ndzip_sharded_fail

// a failure inside libndzip_hip.so: its own thread-local message stands
int forwarded(int status) {
    if (status != NDZIP_HIP_OK) g_error_is_ours = false;
    return status;
}

#define CODEC_TRY(expr)                                  \
    do {                                                 \
        if (int s_ = forwarded(expr)) return s_;         \
    } while (0)

#define HIP_TRY(expr, what)                                                                                      \
    do {                                                                                                         \
        hipError_t e_ = (expr);                                                                                  \
        if (e_ != hipSuccess) return fail(NDZIP_HIP_ERR_RUNTIME, "%s: %s", what, hipGetErrorString(e_));          \
    } while (0)

constexpr uint32_t side_of(int dims) { return dims == 1 ? 4096u : dims == 2 ? 64u : 16u; }  // src/ndzip/common.hh:383-393

size_t word_bytes(int dtype) { return dtype == NDZIP_HIP_F32 ? 4 : 8; }

// shard `rank` of the plan (restated in ndzip_amd/sharded.py: plan_shards -- tests hold the two against each other)
int plan(int dtype, int dims, const uint32_t *extent, uint32_t rank, uint32_t world, ndzip_hip_shard *out) {
    if (!extent || !out) return fail(NDZIP_HIP_ERR_INVALID_ARGUMENT, "null argument");
    if (dtype != NDZIP_HIP_F32 && dtype != NDZIP_HIP_F64) return fail(NDZIP_HIP_ERR_INVALID_ARGUMENT, "invalid dtype");
    if (dims < 1 || dims > 3) return fail(NDZIP_HIP_ERR_INVALID_ARGUMENT, "Invalid dimensionality");
    if (world == 0 || rank >= world) return fail(NDZIP_HIP_ERR_INVALID_ARGUMENT, "rank %u outside a plan of %u", rank, world);
    const uint32_t side = side_of(dims);
    uint64_t per_plane = 1, row_elements = 1;
    bool whole_border = extent[0] / side == 0;
    for (int d = 1; d < dims; ++d) {
        per_plane *= extent[d] / side;
        row_elements *= extent[d];
        whole_border = whole_border || extent[d] / side == 0;
    }
    const uint64_t planes = extent[0] / side;
    const uint64_t p0 = rank * planes / world, p1 = (rank + 1ull) * planes / world;
    ndzip_hip_shard sh{};
    sh.rank = rank;
    sh.world = world;
    sh.start0 = static_cast<uint32_t>(p0 * side);
    const uint64_t stop = rank + 1 < world ? p1 * side : extent[0];
    sh.extent[0] = static_cast<uint32_t>(stop - sh.start0);
    for (int d = 1; d < dims; ++d) sh.extent[d] = extent[d];
    const uint64_t nhc = whole_border ? 0 : (p1 - p0) * per_plane;
    const uint64_t first = whole_border ? 0 : p0 * per_plane;
    if (first + nhc > index_max) return fail(NDZIP_HIP_ERR_LIMIT, "hypercube count does not fit the format's uint32_t");
    sh.hc_begin = static_cast<uint32_t>(first);
    sh.hc_end = static_cast<uint32_t>(first + nhc);
    const uint64_t elements = static_cast<uint64_t>(sh.extent[0]) * row_elements;
    if (elements - nhc * 4096 > index_max) return fail(NDZIP_HIP_ERR_LIMIT, "slab border does not fit the format's uint32_t");
    sh.border_elements = static_cast<uint32_t>(elements - nhc * 4096);
    uint64_t bound = 0;
    uint32_t hw = 0;
    CODEC_TRY(ndzip_hip_compressed_length_bound(dtype, dims, sh.extent, &bound));
    CODEC_TRY(ndzip_hip_header_words(dtype, static_cast<uint32_t>(nhc), &hw));
    sh.body_capacity_words = bound - hw;
    *out = sh;
    return NDZIP_HIP_OK;
}

}  // namespace

// The in-process transport: the ranks are threads, the all-gather a rendezvous + device-to-device copies (see the header).
struct ndzip_hip_local_group {
    std::mutex m;
    std::condition_variable cv;
    uint32_t world = 1, arrived = 0, generation = 0;
    std::vector<const uint32_t *> send;
    void wait() {  // reusable barrier
        std::unique_lock<std::mutex> l(m);
        const uint32_t g = generation;
        if (++arrived == world) {
            arrived = 0;
            ++generation;
            cv.notify_all();
        } else {
            cv.wait(l, [&] { return generation != g; });
        }
    }
};

namespace {
struct local_rank {
    ndzip_hip_local_group *group;
    uint32_t rank;
};
}  // namespace

struct ndzip_hip_sharded {
    std::unique_ptr<local_rank> local;  // create_local: the table's context, owned by the handle
    void *slab_dev = nullptr;            // compress_host / decompress_host: the slab on the device (first use)
    int dtype = 0, dims = 0;
    uint32_t extent[3] = {0, 0, 0};
    uint32_t rank = 0, world = 1;
    hipStream_t stream = nullptr;
    ndzip_hip_collectives coll{};
    ndzip_hip_shard shard{};
    std::vector<ndzip_hip_shard> shards;  // the whole plan
    uint32_t nhc_total = 0, header_words_total = 0;
    uint32_t max_segment = 0;  // most header entries any rank owns
    bool equal_segments = true;
    ndzip_hip_compressor *comp = nullptr;
    ndzip_hip_decompressor *decomp = nullptr;
    // device buffers
    uint32_t *header_local = nullptr;   // max_segment + 1 entries: this rank's, zero-padded to the longest segment
    void *body = nullptr;               // body_capacity_words
    uint32_t *body_len = nullptr;       // words written incl. the slab's border
    uint32_t *base = nullptr;           // global word offset of body[0]
    uint32_t *lens_all = nullptr;       // world entries
    uint32_t *borders = nullptr;        // world entries (constant)
    uint32_t *header_gathered = nullptr; // world x max_segment (the all-gather's landing zone)
    uint32_t *header_global = nullptr;   // nhc_total entries (== header_gathered when the segments are equal)
    bool have_stream = false;            // a compress or a load has filled the handle
    bool exchange_due = false;           // compress_local has run, its exchange has not

    ~ndzip_hip_sharded() {
        if (comp) (void) ndzip_hip_compressor_destroy(comp);
        if (decomp) (void) ndzip_hip_decompressor_destroy(decomp);
        for (void *p : {static_cast<void *>(header_local), body, static_cast<void *>(body_len), static_cast<void *>(base),
                     static_cast<void *>(lens_all), static_cast<void *>(borders), static_cast<void *>(header_gathered)}) {
            if (p) (void) hipFree(p);
        }
        if (header_global && header_global != header_gathered && header_global != header_local) (void) hipFree(header_global);
        if (slab_dev) (void) hipFree(slab_dev);
    }
};

namespace {

int gather(ndzip_hip_sharded *s, const uint32_t *d_send, uint32_t *d_recv, size_t count) {
    const int rc = s->coll.all_gather_u32(s->coll.ctx, d_send, d_recv, count, s->stream);
    if (rc == 0) return NDZIP_HIP_OK;
    const char *txt = s->coll.error_string ? s->coll.error_string(s->coll.ctx, rc) : nullptr;
    return fail(NDZIP_HIP_ERR_RUNTIME, "all-gather of %zu uint32 per rank failed: %s (code %d)", count, txt ? txt : "transport error", rc);
}

// What the stream format can carry at all.  Eight legal slabs can form a global array that is not: the element count and the
// stream length are uint32 (ndzip.hh:20), and so are the header's offsets, which address hypercube runs only
// (common.hh:351-358) -- so the runs are held against the offsets and the whole bound against the length word, separately.
int check_global_extent(int dtype, int dims, const uint32_t *extent, uint32_t nhc_total) {
    uint64_t n = 1;
    for (int d = 0; d < dims; ++d) {  // (three uint32 factors can wrap 64 bits: checked stepwise)
        if (extent[d] != 0 && n > index_max / extent[d]) return fail(NDZIP_HIP_ERR_LIMIT, "global extent has more than 2^32 - 1 elements: index_type (uint32) cannot count them");
        n *= extent[d];
    }
    const uint64_t per_hc = 4096 + 4096 / (word_bytes(dtype) * 8);  // words of an incompressible hypercube (common.cc:31-55)
    if (static_cast<uint64_t>(nhc_total) * per_hc > index_max) {
        return fail(NDZIP_HIP_ERR_LIMIT, "global extent: %u hypercubes can take %llu words, more than the format's 32-bit offsets address", nhc_total,
                static_cast<unsigned long long>(nhc_total * per_hc));
    }
    uint32_t hw = 0;
    CODEC_TRY(ndzip_hip_header_words(dtype, nhc_total, &hw));
    const uint64_t bound = hw + static_cast<uint64_t>(nhc_total) * per_hc + (n - static_cast<uint64_t>(nhc_total) * 4096);
    if (bound > index_max) {
        return fail(NDZIP_HIP_ERR_LIMIT, "global extent: compressed_length_bound = %llu words does not fit the uint32 stream length",
                static_cast<unsigned long long>(bound));
    }
    return NDZIP_HIP_OK;
}

template<typename T>
int device_alloc(T **p, size_t count, const char *what) {
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(p), (count ? count : 1) * sizeof(T)), what);
    return NDZIP_HIP_OK;
}

int create(int dtype, int dims, const uint32_t *global_extent, uint32_t rank, uint32_t world, const ndzip_hip_collectives *coll, void *hip_stream,
        ndzip_hip_sharded **out) {
    if (!out) return fail(NDZIP_HIP_ERR_INVALID_ARGUMENT, "null handle pointer");
    *out = nullptr;
    if (!coll || !coll->all_gather_u32) return fail(NDZIP_HIP_ERR_INVALID_ARGUMENT, "no all_gather_u32 in the collectives table");
    if (ndzip_hip_abi_version() != NDZIP_HIP_ABI_VERSION) {
        return fail(NDZIP_HIP_ERR_RUNTIME, "libndzip_hip.so has ABI version %d, libndzip_hip_rccl.so was built against %d", ndzip_hip_abi_version(), NDZIP_HIP_ABI_VERSION);
    }
    ndzip_hip_shard mine{};
    if (int st = plan(dtype, dims, global_extent, rank, world, &mine)) return st;
    auto *s = new (std::nothrow) ndzip_hip_sharded;
    if (!s) return fail(NDZIP_HIP_ERR_RUNTIME, "out of host memory");
    struct guard {
        ndzip_hip_sharded *p;
        ~guard() { delete p; }
    } g{s};
    s->dtype = dtype;
    s->dims = dims;
    for (int d = 0; d < dims; ++d) s->extent[d] = global_extent[d];
    s->rank = rank;
    s->world = world;
    s->stream = static_cast<hipStream_t>(hip_stream);
    s->coll = *coll;
    s->shard = mine;
    s->shards.resize(world);
    std::vector<uint32_t> borders(world);
    for (uint32_t r = 0; r < world; ++r) {
        if (int st = plan(dtype, dims, global_extent, r, world, &s->shards[r])) return st;
        const uint32_t n = s->shards[r].hc_end - s->shards[r].hc_begin;
        if (n > s->max_segment) s->max_segment = n;
        borders[r] = s->shards[r].border_elements;
    }
    for (uint32_t r = 0; r < world; ++r) s->equal_segments = s->equal_segments && s->shards[r].hc_end - s->shards[r].hc_begin == s->max_segment;
    s->nhc_total = s->shards[world - 1].hc_end;
    if (int st = check_global_extent(dtype, dims, global_extent, s->nhc_total)) return st;
    CODEC_TRY(ndzip_hip_header_words(dtype, s->nhc_total, &s->header_words_total));

    const uint32_t nhc = mine.hc_end - mine.hc_begin;
    CODEC_TRY(ndzip_hip_compressor_create(dtype, dims, nhc, s->stream, &s->comp));
    CODEC_TRY(ndzip_hip_decompressor_create(dtype, dims, s->stream, &s->decomp));
    if (int st = device_alloc(&s->header_local, static_cast<size_t>(s->max_segment) + 1, "header segment")) return st;
    HIP_TRY(hipMalloc(&s->body, (mine.body_capacity_words ? mine.body_capacity_words : 1) * word_bytes(dtype)), "body");
    if (int st = device_alloc(&s->body_len, 1, "body length")) return st;
    if (int st = device_alloc(&s->base, 1, "base")) return st;
    if (int st = device_alloc(&s->lens_all, world, "gathered lengths")) return st;
    if (int st = device_alloc(&s->borders, world, "border counts")) return st;
    // (the padding of a short segment is zeroed ONCE: compress_split writes the rank's own entries only)
    HIP_TRY(hipMemsetAsync(s->header_local, 0, (static_cast<size_t>(s->max_segment) + 1) * sizeof(uint32_t), s->stream), "clearing the header segment");
    HIP_TRY(hipMemsetAsync(s->body_len, 0, sizeof(uint32_t), s->stream), "clearing the body length");
    HIP_TRY(hipMemsetAsync(s->base, 0, sizeof(uint32_t), s->stream), "clearing the base");
    HIP_TRY(hipMemsetAsync(s->lens_all, 0, world * sizeof(uint32_t), s->stream), "clearing the lengths");
    HIP_TRY(hipMemcpyAsync(s->borders, borders.data(), world * sizeof(uint32_t), hipMemcpyHostToDevice, s->stream), "uploading the border counts");
    HIP_TRY(hipStreamSynchronize(s->stream), "uploading the plan");  // (`borders` is a host temporary)
    if (world == 1) {
        s->header_global = s->header_local;  // one shard: local offsets are global offsets, nothing to exchange
    } else {
        if (int st = device_alloc(&s->header_gathered, static_cast<size_t>(world) * s->max_segment, "gathered header")) return st;
        if (s->equal_segments) {
            s->header_global = s->header_gathered;
        } else if (int st = device_alloc(&s->header_global, s->nhc_total, "global header")) {
            return st;
        }
    }
    g.p = nullptr;
    *out = s;
    return NDZIP_HIP_OK;
}

// stream-ordered on the CALLER's stream as the table's contract says: the send buffer is complete when that stream has drained
int local_all_gather_u32(void *ctx, const uint32_t *d_send, uint32_t *d_recv, size_t count, void *hip_stream) {
    auto *c = static_cast<local_rank *>(ctx);
    auto stream = static_cast<hipStream_t>(hip_stream);
    int rc = hipStreamSynchronize(stream) == hipSuccess ? 0 : 1;  // (a failing rank still keeps both rendezvous: nobody is left waiting)
    c->group->send[c->rank] = d_send;
    c->group->wait();
    for (uint32_t r = 0; r < c->group->world && rc == 0; ++r) {
        // (hipMemcpyDefault: the runtime infers both sides from the unified address space -- rank r's buffer may live on another GPU)
        if (hipMemcpyAsync(d_recv + r * count, c->group->send[r], count * sizeof(uint32_t), hipMemcpyDefault, stream) != hipSuccess) rc = 2;
    }
    if (rc == 0 && hipStreamSynchronize(stream) != hipSuccess) rc = 3;
    c->group->wait();  // nobody overwrites its send buffer before everybody has read it
    return rc;
}

const char *local_error_string(void *, int code) {
    return code == 1 ? "the rank's stream did not drain" : code == 2 ? "device-to-device copy failed" : "the copies did not complete";
}

size_t slab_bytes(const ndzip_hip_sharded *s) {
    size_t n = word_bytes(s->dtype);
    for (int d = 0; d < s->dims; ++d) n *= s->shard.extent[d];
    return n;
}

int ensure_slab(ndzip_hip_sharded *s) {
    if (s->slab_dev) return NDZIP_HIP_OK;
    const size_t n = slab_bytes(s);
    HIP_TRY(hipMalloc(&s->slab_dev, n ? n : 1), "slab staging buffer");
    return NDZIP_HIP_OK;
}

}  // namespace

extern "C" {

NDZIP_HIP_API int ndzip_hip_local_group_create(uint32_t world, ndzip_hip_local_group **out) {
    if (!out || world == 0) return fail(NDZIP_HIP_ERR_INVALID_ARGUMENT, "null group pointer or an empty group");
    auto *g = new (std::nothrow) ndzip_hip_local_group;
    if (!g) return fail(NDZIP_HIP_ERR_RUNTIME, "out of host memory");
    g->world = world;
    g->send.assign(world, nullptr);
    *out = g;
    return NDZIP_HIP_OK;
}

NDZIP_HIP_API int ndzip_hip_local_group_destroy(ndzip_hip_local_group *group) {
    delete group;
    return NDZIP_HIP_OK;
}

NDZIP_HIP_API int ndzip_hip_local_group_barrier(ndzip_hip_local_group *group) {
    if (!group) return fail(NDZIP_HIP_ERR_INVALID_ARGUMENT, "null group");
    group->wait();
    return NDZIP_HIP_OK;
}

NDZIP_HIP_API int ndzip_hip_sharded_create_local(int dtype, int dims, const uint32_t *global_extent, uint32_t rank, uint32_t world,
        ndzip_hip_local_group *group, void *hip_stream, ndzip_hip_sharded **out) {
    if (out) *out = nullptr;
    if (!group) return fail(NDZIP_HIP_ERR_INVALID_ARGUMENT, "null group");
    if (group->world != world) return fail(NDZIP_HIP_ERR_INVALID_ARGUMENT, "the group has %u ranks, the plan %u", group->world, world);
    std::unique_ptr<local_rank> ctx(new (std::nothrow) local_rank{group, rank});
    if (!ctx) return fail(NDZIP_HIP_ERR_RUNTIME, "out of host memory");
    const ndzip_hip_collectives table{ctx.get(), local_all_gather_u32, local_error_string};
    if (int st = create(dtype, dims, global_extent, rank, world, &table, hip_stream, out)) return st;
    (*out)->local = std::move(ctx);
    return NDZIP_HIP_OK;
}

NDZIP_HIP_API int ndzip_hip_sharded_device_count(int *count) {
    if (!count) return fail(NDZIP_HIP_ERR_INVALID_ARGUMENT, "null argument");
    *count = 0;
    if (hipGetDeviceCount(count) != hipSuccess || *count <= 0) return fail(NDZIP_HIP_ERR_NO_DEVICE, "no GPU visible (this back-end has no CPU fallback)");
    return NDZIP_HIP_OK;
}

NDZIP_HIP_API int ndzip_hip_sharded_set_device(int device) {
    HIP_TRY(hipSetDevice(device), "hipSetDevice");
    return NDZIP_HIP_OK;
}

NDZIP_HIP_API int ndzip_hip_sharded_compress_local_host(ndzip_hip_sharded *s, const void *host_slab) {
    if (!s) return fail(NDZIP_HIP_ERR_INVALID_ARGUMENT, "null handle");
    const size_t n = slab_bytes(s);
    if (!host_slab && n) return fail(NDZIP_HIP_ERR_INVALID_ARGUMENT, "null slab");
    if (int st = ensure_slab(s)) return st;
    if (n) HIP_TRY(hipMemcpyAsync(s->slab_dev, host_slab, n, hipMemcpyHostToDevice, s->stream), "copying the slab to the device");
    return ndzip_hip_sharded_compress_local(s, s->slab_dev);
}

NDZIP_HIP_API int ndzip_hip_sharded_compress_host(ndzip_hip_sharded *s, const void *host_slab) {
    if (int st = ndzip_hip_sharded_compress_local_host(s, host_slab)) return st;
    return ndzip_hip_sharded_exchange(s);
}

NDZIP_HIP_API int ndzip_hip_sharded_decompress_host(ndzip_hip_sharded *s, void *host_slab) {
    if (!s) return fail(NDZIP_HIP_ERR_INVALID_ARGUMENT, "null handle");
    const size_t n = slab_bytes(s);
    if (!host_slab && n) return fail(NDZIP_HIP_ERR_INVALID_ARGUMENT, "null slab");
    if (int st = ensure_slab(s)) return st;
    if (int st = ndzip_hip_sharded_decompress(s, s->slab_dev)) return st;
    if (n) HIP_TRY(hipMemcpyAsync(host_slab, s->slab_dev, n, hipMemcpyDeviceToHost, s->stream), "copying the slab to the host");
    HIP_TRY(hipStreamSynchronize(s->stream), "copying the slab to the host");
    return NDZIP_HIP_OK;
}

NDZIP_HIP_API int ndzip_hip_sharded_abi_version(void) { return NDZIP_HIP_SHARDED_ABI_VERSION; }

NDZIP_HIP_API const char *ndzip_hip_sharded_last_error(void) { return g_error_is_ours ? g_error : ndzip_hip_last_error(); }

NDZIP_HIP_API int ndzip_hip_sharded_plan(int dtype, int dims, const uint32_t *global_extent, uint32_t rank, uint32_t world, ndzip_hip_shard *out) {
    return plan(dtype, dims, global_extent, rank, world, out);
}

NDZIP_HIP_API int ndzip_hip_sharded_create_with_collectives(int dtype, int dims, const uint32_t *global_extent, uint32_t rank, uint32_t world,
        const ndzip_hip_collectives *collectives, void *hip_stream, ndzip_hip_sharded **out) {
    return create(dtype, dims, global_extent, rank, world, collectives, hip_stream, out);
}

NDZIP_HIP_API int ndzip_hip_sharded_shard(const ndzip_hip_sharded *s, ndzip_hip_shard *out) {
    if (!s || !out) return fail(NDZIP_HIP_ERR_INVALID_ARGUMENT, "null argument");
    *out = s->shard;
    return NDZIP_HIP_OK;
}

NDZIP_HIP_API int ndzip_hip_sharded_compress_local(ndzip_hip_sharded *s, const void *d_in_slab) {
    if (!s) return fail(NDZIP_HIP_ERR_INVALID_ARGUMENT, "null handle");
    // (1) the slab, offsets local to this rank's body
    CODEC_TRY(ndzip_hip_compressor_compress_split(s->comp, d_in_slab, s->dims, s->shard.extent, s->header_local, s->body, s->body_len));
    s->have_stream = true;
    s->exchange_due = true;
    return NDZIP_HIP_OK;
}

NDZIP_HIP_API int ndzip_hip_sharded_exchange(ndzip_hip_sharded *s) {
    if (!s) return fail(NDZIP_HIP_ERR_INVALID_ARGUMENT, "null handle");
    // (adding the base twice would corrupt the entries: one exchange per compress_local)
    if (!s->exchange_due) return fail(NDZIP_HIP_ERR_INVALID_ARGUMENT, "exchange without a compress_local before it");
    s->exchange_due = false;
    const uint32_t nhc = s->shard.hc_end - s->shard.hc_begin;
    if (s->world == 1) {
        // the base of the only shard is 0 (already there); the error-word rule of step (3) still applies to a single slab
        return forwarded(ndzip_hip_compressor_offset_header_gathered(s->comp, s->header_local, 0, s->body_len, s->borders, 0, 1, s->base));
    }
    // (2) one uint32 per rank
    if (int st = gather(s, s->body_len, s->lens_all, 1)) return st;
    // (3) base from ALL lengths (64-bit sums, overflow -> error word on every rank), added to this rank's entries
    CODEC_TRY(ndzip_hip_compressor_offset_header_gathered(s->comp, s->header_local, nhc, s->lens_all, s->borders, s->rank, s->world, s->base));
    // (4) the header segments, padded to the longest (the padding is zero and is dropped again below)
    if (s->max_segment == 0) return NDZIP_HIP_OK;
    if (int st = gather(s, s->header_local, s->header_gathered, s->max_segment)) return st;
    if (!s->equal_segments) {
        for (uint32_t r = 0; r < s->world; ++r) {
            const uint32_t n = s->shards[r].hc_end - s->shards[r].hc_begin;
            if (n == 0) continue;
            HIP_TRY(hipMemcpyAsync(s->header_global + s->shards[r].hc_begin, s->header_gathered + static_cast<size_t>(r) * s->max_segment,
                            n * sizeof(uint32_t), hipMemcpyDeviceToDevice, s->stream),
                    "compacting the gathered header");
        }
    }
    return NDZIP_HIP_OK;
}

NDZIP_HIP_API int ndzip_hip_sharded_compress(ndzip_hip_sharded *s, const void *d_in_slab) {
    if (int st = ndzip_hip_sharded_compress_local(s, d_in_slab)) return st;
    return ndzip_hip_sharded_exchange(s);
}

NDZIP_HIP_API int ndzip_hip_sharded_decompress(ndzip_hip_sharded *s, void *d_out_slab) {
    if (!s) return fail(NDZIP_HIP_ERR_INVALID_ARGUMENT, "null handle");
    if (!s->have_stream) return fail(NDZIP_HIP_ERR_INVALID_ARGUMENT, "nothing to decode: neither compress nor load has run on this handle");
    if (s->exchange_due) {
        // the slab decodes from its LOCAL offsets just as well (base 0): nothing from the other ranks is needed, so a host may put
        // the decode in front of the exchange
        return forwarded(ndzip_hip_decompressor_decompress_split_bounded(s->decomp, s->header_local, nullptr, s->body,
                static_cast<uint32_t>(s->shard.body_capacity_words), d_out_slab, s->dims, s->shard.extent));
    }
    // this rank's entries with global offsets == header_global[hc_begin, hc_end); its base == the entry in front of them
    return forwarded(ndzip_hip_decompressor_decompress_split_bounded(s->decomp, s->header_local, s->base, s->body,
            static_cast<uint32_t>(s->shard.body_capacity_words), d_out_slab, s->dims, s->shard.extent));
}

NDZIP_HIP_API int ndzip_hip_sharded_header_global(const ndzip_hip_sharded *s, const uint32_t **d_header, uint32_t *num_entries) {
    if (!s) return fail(NDZIP_HIP_ERR_INVALID_ARGUMENT, "null handle");
    if (d_header) *d_header = s->header_global;
    if (num_entries) *num_entries = s->nhc_total;
    return NDZIP_HIP_OK;
}

NDZIP_HIP_API int ndzip_hip_sharded_body(const ndzip_hip_sharded *s, const void **d_body, const uint32_t **d_body_length_words,
        const uint32_t **d_base_words) {
    if (!s) return fail(NDZIP_HIP_ERR_INVALID_ARGUMENT, "null handle");
    if (d_body) *d_body = s->body;
    if (d_body_length_words) *d_body_length_words = s->body_len;
    if (d_base_words) *d_base_words = s->base;
    return NDZIP_HIP_OK;
}

NDZIP_HIP_API int ndzip_hip_sharded_check(ndzip_hip_sharded *s) {
    if (!s) return fail(NDZIP_HIP_ERR_INVALID_ARGUMENT, "null handle");
    const int a = ndzip_hip_compressor_check(s->comp);  // (both are read and cleared even when the first one reports)
    if (a != NDZIP_HIP_OK) {
        snprintf(g_error, sizeof g_error, "%s", ndzip_hip_last_error());
        (void) ndzip_hip_decompressor_check(s->decomp);
        g_error_is_ours = true;
        return a;
    }
    return forwarded(ndzip_hip_decompressor_check(s->decomp));
}

NDZIP_HIP_API int ndzip_hip_sharded_stream_layout(ndzip_hip_sharded *s, ndzip_hip_stream_layout *out) {
    if (!s || !out) return fail(NDZIP_HIP_ERR_INVALID_ARGUMENT, "null argument");
    if (!s->have_stream) return fail(NDZIP_HIP_ERR_INVALID_ARGUMENT, "no stream: neither compress nor load has run on this handle");
    if (s->exchange_due) return fail(NDZIP_HIP_ERR_INVALID_ARGUMENT, "the offsets are still local: ndzip_hip_sharded_exchange has not run since compress_local");
    if (int st = ndzip_hip_sharded_check(s)) return st;
    std::vector<uint32_t> lens(s->world);
    if (s->world == 1) {
        HIP_TRY(hipMemcpyAsync(lens.data(), s->body_len, sizeof(uint32_t), hipMemcpyDeviceToHost, s->stream), "reading the body length");
    } else {
        HIP_TRY(hipMemcpyAsync(lens.data(), s->lens_all, s->world * sizeof(uint32_t), hipMemcpyDeviceToHost, s->stream), "reading the gathered lengths");
    }
    HIP_TRY(hipStreamSynchronize(s->stream), "reading the gathered lengths");
    uint64_t runs_before = 0, runs_total = 0, border_before = 0, border_total = 0;
    for (uint32_t r = 0; r < s->world; ++r) {
        const uint32_t b = s->shards[r].border_elements;
        if (lens[r] < b) return fail(NDZIP_HIP_ERR_DEVICE_FAULT, "rank %u reports %u words, fewer than its %u border words (a failed launch stores 0)", r, lens[r], b);
        if (r == s->rank) {
            runs_before = runs_total;
            border_before = border_total;
        }
        runs_total += lens[r] - b;
        border_total += b;
    }
    ndzip_hip_stream_layout l{};
    l.header_words = s->header_words_total;
    l.runs_offset_words = l.header_words + runs_before;
    l.runs_words = lens[s->rank] - s->shard.border_elements;
    l.border_offset_words = l.header_words + runs_total + border_before;
    l.border_words = s->shard.border_elements;
    l.stream_words = l.header_words + runs_total + border_total;
    *out = l;
    return NDZIP_HIP_OK;
}

NDZIP_HIP_API int ndzip_hip_sharded_write_stream(ndzip_hip_sharded *s, void *host_stream, uint64_t capacity_words, int with_header) {
    if (!s || !host_stream) return fail(NDZIP_HIP_ERR_INVALID_ARGUMENT, "null argument");
    ndzip_hip_stream_layout l{};
    if (int st = ndzip_hip_sharded_stream_layout(s, &l)) return st;
    if (l.stream_words > capacity_words) {
        return fail(NDZIP_HIP_ERR_CAPACITY, "the stream has %llu words, the buffer %llu", static_cast<unsigned long long>(l.stream_words),
                static_cast<unsigned long long>(capacity_words));
    }
    const size_t wb = word_bytes(s->dtype);
    char *dst = static_cast<char *>(host_stream);
    const char *body = static_cast<const char *>(s->body);
    if (with_header) {
        // header_words whole words of the dtype; for 64-bit words an odd entry count leaves half a word, which the format zeroes
        memset(dst + (l.header_words ? (l.header_words - 1) * wb : 0), 0, l.header_words ? wb : 0);
        if (s->nhc_total) {
            HIP_TRY(hipMemcpyAsync(dst, s->header_global, static_cast<size_t>(s->nhc_total) * sizeof(uint32_t), hipMemcpyDeviceToHost, s->stream), "copying the header");
        }
    }
    if (l.runs_words) HIP_TRY(hipMemcpyAsync(dst + l.runs_offset_words * wb, body, l.runs_words * wb, hipMemcpyDeviceToHost, s->stream), "copying the runs");
    if (l.border_words) {
        HIP_TRY(hipMemcpyAsync(dst + l.border_offset_words * wb, body + l.runs_words * wb, l.border_words * wb, hipMemcpyDeviceToHost, s->stream), "copying the border");
    }
    HIP_TRY(hipStreamSynchronize(s->stream), "copying the stream");
    return NDZIP_HIP_OK;
}

NDZIP_HIP_API int ndzip_hip_sharded_load(ndzip_hip_sharded *s, const void *host_stream, uint64_t stream_words) {
    if (!s || !host_stream) return fail(NDZIP_HIP_ERR_INVALID_ARGUMENT, "null argument");
    uint32_t words = 0;
    // validates every header entry against its predecessor and the implied stream against stream_words
    CODEC_TRY(ndzip_hip_stream_words(s->dtype, s->dims, s->extent, host_stream, stream_words, &words));
    const size_t wb = word_bytes(s->dtype);
    const char *src = static_cast<const char *>(host_stream);
    const uint32_t *header = static_cast<const uint32_t *>(host_stream);
    const uint32_t nhc = s->shard.hc_end - s->shard.hc_begin;
    const uint32_t base = s->shard.hc_begin ? header[s->shard.hc_begin - 1] : 0;
    const uint32_t end = nhc ? header[s->shard.hc_end - 1] : base;
    const uint64_t runs_total = s->nhc_total ? header[s->nhc_total - 1] : 0;
    uint64_t border_before = 0;
    for (uint32_t r = 0; r < s->rank; ++r) border_before += s->shards[r].border_elements;
    const uint64_t runs_words = end - base, border_words = s->shard.border_elements;
    if (runs_words + border_words > s->shard.body_capacity_words) return fail(NDZIP_HIP_ERR_INVALID_ARGUMENT, "stream pieces of this rank exceed its body bound");
    const uint32_t len = static_cast<uint32_t>(runs_words + border_words);
    char *body = static_cast<char *>(s->body);
    if (nhc) HIP_TRY(hipMemcpyAsync(s->header_local, header + s->shard.hc_begin, nhc * sizeof(uint32_t), hipMemcpyHostToDevice, s->stream), "uploading the header entries");
    if (runs_words) HIP_TRY(hipMemcpyAsync(body, src + (s->header_words_total + base) * wb, runs_words * wb, hipMemcpyHostToDevice, s->stream), "uploading the runs");
    if (border_words) {
        HIP_TRY(hipMemcpyAsync(body + runs_words * wb, src + (s->header_words_total + runs_total + border_before) * wb, border_words * wb, hipMemcpyHostToDevice, s->stream),
                "uploading the border");
    }
    // the lengths every rank's compress would have gathered, from the header alone (ndzip_hip_sharded_stream_layout reads them)
    std::vector<uint32_t> lens(s->world);
    for (uint32_t r = 0; r < s->world; ++r) {
        const ndzip_hip_shard &q = s->shards[r];
        const uint32_t b = q.hc_begin ? header[q.hc_begin - 1] : 0, e = q.hc_end > q.hc_begin ? header[q.hc_end - 1] : b;
        lens[r] = e - b + q.border_elements;
    }
    HIP_TRY(hipMemcpyAsync(s->lens_all, lens.data(), s->world * sizeof(uint32_t), hipMemcpyHostToDevice, s->stream), "uploading the lengths");
    HIP_TRY(hipMemcpyAsync(s->base, &base, sizeof base, hipMemcpyHostToDevice, s->stream), "uploading the base");
    HIP_TRY(hipMemcpyAsync(s->body_len, &len, sizeof len, hipMemcpyHostToDevice, s->stream), "uploading the body length");
    s->exchange_due = false;
    HIP_TRY(hipStreamSynchronize(s->stream), "uploading the stream pieces");  // (`base`, `len`, `lens` are host temporaries)
    s->have_stream = true;
    return NDZIP_HIP_OK;
}

NDZIP_HIP_API int ndzip_hip_sharded_destroy(ndzip_hip_sharded *s) {
    if (!s) return NDZIP_HIP_OK;
    (void) hipStreamSynchronize(s->stream);
    delete s;
    return NDZIP_HIP_OK;
}

}  // extern "C"
