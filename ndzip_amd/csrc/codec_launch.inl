// ndzip_amd/csrc/codec_launch.inl -- __global__ kernels and launchers, instantiated for one value type
// (NDZIP_T) per translation unit so the two types compile in parallel.
//
// Kernels (reference counterparts in src/ndzip/cuda_codec.inl, none of whose structure is reused):
//   compress_kernel    persistent, one tile of K hypercubes per workgroup iteration: encode in LDS, publish the
//                      tile length, decoupled look-back for the tile's stream offset, coalesced copy-out, header
//                      entries, stream length.  Replaces compress_block + hierarchical_inclusive_scan (2 kernels x
//                      levels) + compact_all_chunks + store_stream_length (:401-457,:507-511, cuda_bits.cuh:266-333).
//   decompress_kernel  one tile per workgroup, header lookup -> LDS -> decode (replaces decompress_block :477-492).
//   debug_*_kernel     single-hypercube stage entry points for the parity tests: compiled only with NDZIP_STAGE_KERNELS, i.e. into
//                      libndzip_hip_stages.so (stages_f32.hip / stages_f64.hip) -- the product library holds none of them.

#include <cstdio>
#include <cstdlib>

#include "codec_kernels.hpp"
#include "codec_kernels_wide.hpp"
#include "codec_launch.hpp"

#ifndef NDZIP_T
#error "define NDZIP_T before including codec_launch.inl"
#endif

namespace ndzip_hip {

namespace {

using T_ = NDZIP_T;

// hypercubes per workgroup: f32 pairs two hypercubes (x-adjacent in 3D, so the pair covers whole 128-byte
// lines of the 64-byte cube rows); f64 rows already are >= 128 bytes and the cube is twice as large in LDS.
template<typename T, int Dims>
struct tile_cfg {
    static constexpr int K = sizeof(T) == 4 ? 2 : 1;
    static constexpr int threads = K * threads_per_hc;
    using W = typename word_of<T>::type;
    using L = lds_layout<W>;
    static constexpr uint32_t xchg_bytes = 32;  // per hypercube: 2 x uint32 + 2 x W
    // bytes between the staging regions of a tile's hypercubes: = 64 mod 128, see stage_pair_regs
    static constexpr uint32_t cube_stride = L::cube_bytes + (K > 1 ? 64 : 0);
    static constexpr uint32_t smem_bytes = K * cube_stride + L::zero_bytes + K * xchg_bytes + 32;
};

// Descriptor = (status << 32) | value with status = (epoch << 2) | state.  The epoch is a per-scratch launch counter (kept IN the
// scratch and advanced by the kernel itself, codec_launch.hpp: epoch_word): a
// descriptor left behind by an earlier launch has another epoch and reads as "not published", so the scratch needs no
// clearing between launches (a 130 KiB memset node in front of every compress call cost ~4 us of a 200 us launch, and
// 10 % of a 16 Mi-element one).  States: 0 unpublished, 1 aggregate, 2 inclusive prefix, 3 lane outside the window.
struct desc_ref {
    tile_desc *p;
    uint32_t epoch;
};
NDZIP_DEV tile_desc desc_tag(uint32_t epoch, uint32_t state) {
    return static_cast<tile_desc>((epoch << 2) | state) << 32;
}
NDZIP_DEV uint32_t desc_state(tile_desc d, uint32_t epoch) {
    const uint32_t status = static_cast<uint32_t>(d >> 32);
    return (status >> 2) == epoch ? (status & 3u) : 0u;
}
// Look-back window: how many predecessor descriptors one hop reads (one per lane).  Every descriptor read is an
// uncached 8-byte agent-scope load, i.e. its own fabric transaction: measured on 512^3 f32, 256 per hop costs 45 us
// more kernel time than 64 per hop, 1024 per hop 170 us more (profiles/) -- narrow windows win.
constexpr int lookback_lanes = 64;
// Polls of a missing predecessor's descriptor after which a look-back gives up and sets the error word.  One poll is an
// uncached agent-scope load round trip (>= 1 us under load) plus s_sleep 8, so 2^20 of them are a second or more -- against the
// few microseconds a predecessor of a fully resident grid takes to publish.  A count, not a clock reading, on purpose: a time
// budget (s_memrealtime) was tried in round 3 and its 64-bit scalars in the -- never executed -- waiting path cost the hot loop
// registers: the compress kernels sit exactly at 128 VGPRs, and five of them went to scratch.  Overridable only so that the
// parity tests can force the give-up path (0 = any wait for a predecessor is a time-out).
#ifndef NDZIP_LOOKBACK_SPIN_LIMIT
#define NDZIP_LOOKBACK_SPIN_LIMIT (1u << 20)
#endif
constexpr uint32_t spin_limit = NDZIP_LOOKBACK_SPIN_LIMIT;

NDZIP_DEV tile_desc desc_load(const tile_desc *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
NDZIP_DEV void desc_store(tile_desc *p, tile_desc v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// One look-back window: lane l reads the descriptor of tile `nearest - l` (nearest = the window's first, i.e. highest, tile;
// wave-uniform) -- "inclusive prefix 0" in front of tile 0, "outside the window" for lanes beyond it.  The address is a
// wave-uniform base (the descriptor of tile nearest - 63, possibly in front of the array: it is only ever added to) plus a
// per-lane byte offset, so that the load takes the SGPR-base form and no lane carries a 64-bit descriptor pointer around the loop.
NDZIP_DEV tile_desc window_load(desc_ref desc, long long nearest, int lane) {
    const long long idx = nearest - lane;
    if (lane >= lookback_lanes) return desc_tag(desc.epoch, 3u);
    if (idx < 0) return desc_tag(desc.epoch, 2u);
    const char *base = reinterpret_cast<const char *>(desc.p) + (nearest - 63) * static_cast<long long>(sizeof(tile_desc));
    const uint32_t off = lane_offset_here(static_cast<uint32_t>(63 - lane) * static_cast<uint32_t>(sizeof(tile_desc)));
    return desc_load(reinterpret_cast<const tile_desc *>(scalar_pointer(base) + off));
}

// Ticket n of class c -> tile: plain interleave.  A class's tiles increase with its tickets.
NDZIP_DEV uint32_t tile_of_ticket(uint32_t ticket, uint32_t cls, uint32_t num_classes) { return ticket * num_classes + cls; }

// tickets[class * ticket_stride_words] = next ticket of the class; tickets[ticket_classes * ticket_stride_words] =
// number of workgroups that have drawn their last ticket
// The last workgroup to leave also looks at the error word: a launch that hit a look-back timeout has published offsets
// that are too small, so its stream is garbage although every write stayed in bounds -- the stream length is then
// poisoned to 0 (shorter than any valid stream: every consumer of the length fails loudly, ndzip_hip_stream_words and
// the decompress entry points reject it) for callers that never call ndzip_hip_compressor_check().  Ordering without a
// cache write-back: work-item 0 of a workgroup is the only one that sets the error word, stores the stream length and
// increments `done`, and the first two are RETURNING agent-scope atomics (gfx950_lds.hpp: *_performed) -- they have been
// performed where all XCDs meet before the `done` increment is issued, so the workgroup that reads `done == grid - 1` and then
// the error word (agent-scope load) sees both, and its poison store comes after every length store.  (An agent-scope release
// fence here = a write-back of the XCD's whole L2, ~3.5 us per workgroup: MI355X guide.)
NDZIP_DEV void store_stream_length(uint32_t *out_len, uint32_t words) {
    if (out_len) exchange_performed(out_len, words);
}
// The launch's epoch: one agent-scope load per work-item on the way in (the word was written by the previous launch's last workgroup,
// a kernel boundary ago).  In two halves so that the load's round trip -- a cold line: it comes from memory -- is not waited for in
// front of the first ticket: issued first of all, turned into a scalar (v_readfirstlane, i.e. the s_waitcnt) only behind the first
// barrier, by which time wavefront 0 has had its ticket back (loads and atomics return in order) and the others have been waiting.
NDZIP_DEV uint32_t launch_epoch_load(const uint32_t *tickets) {
    return __hip_atomic_load(tickets + epoch_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
NDZIP_DEV uint32_t launch_epoch(uint32_t loaded) { return static_cast<uint32_t>(wave_uniform(static_cast<int>(loaded))); }
// (`tid`: work-item id in a one-dimensional workgroup of WorkItems; EVERY work-item of the workgroup calls this, from converged code:
// wavefront 0 reads lane 0's verdict with a shuffle and wipes in strides of 64, so all of its 64 lanes have to be there.  The
// workgroup size is a template argument so that the contract is checked where a kernel instantiates it, not in a comment.)
template<int WorkItems>
NDZIP_DEV void release_tickets(uint32_t *tickets, uint32_t num_classes, int tid, uint32_t *err, uint32_t *out_len, desc_ref desc) {
    static_assert(WorkItems >= 64 && WorkItems % 64 == 0, "release_tickets: wavefront 0 must be a full wavefront (whole wavefronts per workgroup)");
    if (tid >= 64) return;  // wavefront 0 stays together: its 64 lanes share the wipe below, work-item 0 does everything else
    uint32_t *done = tickets + ticket_classes * ticket_stride_words;
    uint32_t last = 0;
    if (tid == 0) {
        wait_for_own_memory_operations();
        last = atomicAdd(done, 1u) == gridDim.x - 1 ? 1u : 0u;
    }
    if (__shfl(last, 0, 64) == 0) return;
    if (tid == 0) {
        for (uint32_t c = 0; c < num_classes; ++c) tickets[c * ticket_stride_words] = 0;
        *done = 0;
        if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) store_stream_length(out_len, 0u);
    }
    // Every other workgroup has left: nobody reads the epoch or a descriptor of this launch any more.  The next launch on this
    // scratch runs under epoch + 1; when the 30-bit field starts over, a descriptor an old launch left behind could carry the
    // new epoch again, so the scratch is wiped first (once in 2^30 launches; 64 descriptors per step, with the descriptors' own
    // write-through stores -- up to ~10^6 of them: one work-item alone would hold the launch's end for milliseconds).  The next
    // launch is a kernel boundary away: it sees the wipe and the epoch however the two are ordered here.
    uint32_t next = desc.epoch + 1;
    if (next >= epoch_limit) {
        const uint32_t count = tickets[epoch_word + 1];
        for (uint32_t i = static_cast<uint32_t>(tid); i < count; i += 64) desc_store(desc.p + i, 0);
        next = 1;
    }
    if (tid == 0) __hip_atomic_store(tickets + epoch_word, next, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- decoupled look-back over tile lengths ---------------------------------------------------------------------
// A tile publishes its length (aggregate) as soon as its chunk scan is done, keeps working, and only later
// resolves its exclusive prefix by walking back over the predecessors (nearest first, 256 per hop) until a tile
// with a known inclusive prefix is found.  Forward progress: the grid is persistent and fully resident and every
// workgroup handles its tiles in increasing order, so the smallest unfinished tile never waits.  Every spin is
// bounded; on timeout the error word is set and a partial (smaller) sum of PUBLISHED lengths is returned (lanes
// whose descriptor is unpublished contribute nothing), which keeps all writes inside the caller's buffer.  While a predecessor is missing only ONE lane polls ONE descriptor (with s_sleep):
// window-wide polling by a thousand workgroups would eat the memory system (MI355X guide, "polling-cost").

NDZIP_DEV void publish_aggregate(desc_ref desc, uint32_t tile, uint32_t aggregate) {
    desc_store(desc.p + tile, desc_tag(desc.epoch, tile == 0 ? 2u : 1u) | aggregate);
}

// first look-back window of `tile`, issued early so its latency hides behind other work (vector loads return in
// order: issue this BEFORE any bulk load of the same wavefront)
constexpr int lookback_prefetch = 1;  // windows read ahead of time (asynchronously)

struct lookback_windows {
    tile_desc d[lookback_prefetch];
};

NDZIP_DEV void lookback_issue(desc_ref desc, uint32_t tile, int lane, lookback_windows &w) {
#pragma unroll
    for (int j = 0; j < lookback_prefetch; ++j) {
        w.d[j] = window_load(desc, static_cast<long long>(tile) - 1 - j * lookback_lanes, lane);
    }
}

template<bool Preloaded>
NDZIP_DEV uint32_t resolve_exclusive_prefix_impl(desc_ref desc, uint32_t tile, uint32_t aggregate, uint32_t *err, int lane,
        const lookback_windows &pre) {
    if (tile == 0) return 0;
    if constexpr (Preloaded) {
        // Fast path, no memory operation and therefore no s_waitcnt: the preloaded windows reach an inclusive prefix
        // and every nearer descriptor is published.  (gfx9 counts loads and stores in one vmcnt and hipcc waits
        // vmcnt(0) around the general loop below -- with this wavefront's aggregate store or prefetch loads in flight
        // that wait costs a full memory round trip, 6-9k cycles per iteration, measured.)
        uint32_t sum = 0;
        bool complete = true, done = false;
#pragma unroll
        for (int j = 0; j < lookback_prefetch; ++j) {
            const tile_desc d = pre.d[j];
            const uint32_t status = desc_state(d, desc.epoch);
            const unsigned long long invalid = __ballot(status == 0);
            const unsigned long long inclusive = __ballot(status == 2);
            if (complete && !done) {
                if (inclusive != 0) {
                    const int lf = __builtin_ctzll(inclusive);
                    const unsigned long long nearer = lf == 0 ? 0ull : (~0ull >> (64 - lf));
                    if ((invalid & nearer) == 0) {
                        sum += wave_sum(lane <= lf ? static_cast<uint32_t>(d) : 0u);
                        done = true;
                    } else {
                        complete = false;
                    }
                } else if (invalid == 0) {
                    sum += wave_sum(static_cast<uint32_t>(d));  // (skip lanes carry value 0)
                } else {
                    complete = false;
                }
            }
        }
        if (done) {
            if (lane == 0) desc_store(desc.p + tile, desc_tag(desc.epoch, 2u) | (sum + aggregate));
            return sum;
        }
    }
    uint32_t exclusive = 0;
    long long base = static_cast<long long>(tile) - 1;
    bool timed_out = false;
    uint32_t spins = 0;
    int hop = 0;
    for (;;) {
        bool found = false;
        int lf = 0;
        bool use_preloaded = Preloaded && hop < lookback_prefetch;
        tile_desc d = 0;
        if (use_preloaded) {
#pragma unroll
            for (int j = 0; j < lookback_prefetch; ++j) {
                if (j == hop) d = pre.d[j];
            }
        }
        for (;;) {
            if (!use_preloaded) d = window_load(desc, base, lane);
            use_preloaded = false;
            const uint32_t status = desc_state(d, desc.epoch);
            const unsigned long long invalid = __ballot(status == 0);
            const unsigned long long inclusive = __ballot(status == 2);
            found = inclusive != 0;
            int wait_pos = -1;
            if (found) {
                lf = __builtin_ctzll(inclusive);
                const unsigned long long nearer = lf == 0 ? 0ull : (~0ull >> (64 - lf));
                if (invalid & nearer) wait_pos = __builtin_ctzll(invalid & nearer);
            } else if (invalid != 0) {
                wait_pos = __builtin_ctzll(invalid);
            }
            if (wait_pos < 0) break;
            // the nearest missing predecessor: one lane polls it, then the window is read again
            if (lane == 0) {
                const tile_desc *p = desc.p + (base - wait_pos);
                while (desc_state(desc_load(p), desc.epoch) == 0 && spins < spin_limit) {
                    __builtin_amdgcn_s_sleep(8);
                    ++spins;
                }
            }
            spins = __shfl(spins, 0, 64);
            if (spins >= spin_limit) {
                timed_out = true;  // (`found` / `lf` stay: an inclusive prefix in the window still bounds what may be summed)
                break;
            }
        }
        // After a timeout some lanes still hold UNPUBLISHED descriptors, whose value field is whatever an earlier launch
        // left there (the scratch is never cleared: epoch tags): those lanes contribute 0.  Lanes beyond the nearest
        // inclusive prefix stay excluded as always (their lengths are part of that prefix).  So the prefix returned on a
        // timeout is a partial sum of real lengths -- never larger than the true prefix, which keeps every write of this
        // tile inside the caller's buffer.
        const bool take = (!found || lane <= lf) && desc_state(d, desc.epoch) != 0;
        exclusive += wave_sum(take ? static_cast<uint32_t>(d) : 0u);
        if (found || timed_out) break;
        base -= lookback_lanes;
        ++hop;
    }
    if (timed_out && lane == 0) fetch_or_performed(err, 1u);
    if (lane == 0) desc_store(desc.p + tile, desc_tag(desc.epoch, 2u) | (exclusive + aggregate));
    return exclusive;
}

NDZIP_DEV uint32_t resolve_exclusive_prefix(desc_ref desc, uint32_t tile, uint32_t aggregate, uint32_t *err, int lane) {
    lookback_windows none{};
    return resolve_exclusive_prefix_impl<false>(desc, tile, aggregate, err, lane, none);
}

// Coalesced copy of `n` words from LDS (16-byte aligned `src`) to global: scalar head up to the first 16-byte
// boundary of the destination, 16-byte vector stores, scalar tail.  The destination's alignment relative to the
// LDS run is arbitrary (0..3 uint32), so every output vector straddles two aligned LDS vectors: both are read as
// 16 bytes per lane (conflict-free across lanes; four ds_read_b32 at a 16-byte lane stride would be a 4-way bank
// conflict) and the straddle is resolved by a wave-uniform switch.  Reads up to 16 bytes past the run (inside LDS).
template<int S>
NDZIP_DEV vec16 straddle(const vec16 &lo, const vec16 &hi) {
    vec16 x;
#pragma unroll
    for (int j = 0; j < 4; ++j) x.w[j] = S + j < 4 ? lo.w[(S + j) & 3] : hi.w[(S + j) & 3];
    return x;
}

// `d16`: wave-uniform (scalar_pointer); the per-lane part of every store address is the 32-bit 16 * v
template<typename R, int S, int Threads>
NDZIP_DEV void copy_vectors(const vec16 *__restrict__ a, char *d16, uint32_t nvec, int tid) {
    const char *base = reinterpret_cast<const char *>(a);
    for (uint32_t v = tid; v < nvec; v += Threads) {
        const vec16 lo = lds_read16(R::ptr(base + 16 * v));
        vec16 *d = reinterpret_cast<vec16 *>(d16 + 16u * v);
        if constexpr (S == 0) {
            *d = lo;
        } else {
            *d = straddle<S>(lo, lds_read16(R::ptr(base + 16 * v + 16)));
        }
    }
}

// `dst` must be the same in every lane of a wavefront (a tile's place in the stream): it is kept in scalar registers, and the
// alignment case analysis below is scalar code.
template<typename W, int Threads>
NDZIP_DEV void copy_out(const W *__restrict__ src, W *dst_any, uint32_t n, int tid) {
    constexpr uint32_t wpv = 16 / sizeof(W);
    W *dst = scalar_pointer(dst_any);
    uint32_t lead = (wpv - static_cast<uint32_t>((reinterpret_cast<uintptr_t>(dst) / sizeof(W)) % wpv)) % wpv;
    if (lead > n) lead = n;
    using R = run_layout<W>;  // (`src` is the start of the run's region)
    const auto word = [&](uint32_t i) { return *R::ptr(src + i); };
    const uint32_t t_bytes = static_cast<uint32_t>(tid) * static_cast<uint32_t>(sizeof(W));
    if (static_cast<uint32_t>(tid) < lead) *reinterpret_cast<W *>(reinterpret_cast<char *>(dst) + lane_offset_here(t_bytes)) = word(tid);
    const uint32_t nvec = (n - lead) / wpv;
    const vec16 *a = reinterpret_cast<const vec16 *>(src);
    char *d16 = reinterpret_cast<char *>(scalar_pointer(dst + lead));
    switch (lead * (sizeof(W) / 4)) {  // uint32 offset of the first vector inside its aligned LDS vector
        case 0: copy_vectors<R, 0, Threads>(a, d16, nvec, tid); break;
        case 1: copy_vectors<R, 1, Threads>(a, d16, nvec, tid); break;
        case 2: copy_vectors<R, 2, Threads>(a, d16, nvec, tid); break;
        default: copy_vectors<R, 3, Threads>(a, d16, nvec, tid); break;
    }
    const uint32_t done = lead + nvec * wpv;
    if (static_cast<uint32_t>(tid) < n - done) {
        *reinterpret_cast<W *>(reinterpret_cast<char *>(scalar_pointer(dst + done)) + lane_offset_here(t_bytes)) = word(done + tid);
    }
}

// Tiles are handed out dynamically: `num_classes` ticket counters (class = blockIdx % num_classes, ticket n of class c
// is tile n * num_classes + c), so tile order == start order, a tile's predecessors were all started before it, and
// the look-back never depends on co-residency, dispatch order or placement.
// One tile per ticket: handing a workgroup several CONSECUTIVE tiles (so that only the first needs the look-back) was
// tried and serialises the grid -- the first tile of ticket q then waits for the LAST tile of ticket q-1, which its owner
// reaches iterations later (2 tiles per ticket: 0.215 vs 0.201 ms; 4: 32 ms of bounded spinning).
// Several counters, EACH IN ITS OWN CACHE LINE: returning atomics on one line serialise in its L2 channel at ~80-90 per
// microsecond in total -- no matter how many words of the line they target -- and a 512^3 grid wants ~150 tickets per
// microsecond.  With 16 counters packed into one line the ticket rate capped the whole kernel (tools/membench2.hip:
// the bare load loop 0.211 ms vs 0.101 ms with the counters 64+ bytes apart; static assignment 0.114 ms).

// ---- deferred write-out variant (f32) --------------------------------------------------------------------------------
// A tile is written out ONE ITERATION after it was encoded:
//   * its look-back window is read asynchronously at the top of the next iteration (before that wavefront's prefetch
//     loads: vector loads return in order) and consumed after that iteration's stencil -- an in-launch hand-off on
//     this part costs about as long as the reading CU's memory queue takes to drain (several microseconds under a
//     streaming load), which is now hidden, and the predecessors have had a whole iteration to publish;
//   * the next tile's input is prefetched a whole iteration ahead, so HBM stays busy during compute.
// The encoded tile waits in REGISTERS (its 32 transposed planes per work-item), not in a second LDS buffer: the LDS
// footprint stays at one staging region per hypercube (37 KiB per workgroup), which with 128 VGPRs admits 4 workgroups =
// 16 wavefronts per CU (rounds 1-2: 168 VGPRs, 3 workgroups; a second LDS buffer allowed only 2: measured 0.27 vs 0.32 ms).
template<typename T, int Dims, bool Paired>
struct db_cfg {
    using C = tile_cfg<T, Dims>;
    static constexpr uint32_t smem_bytes = C::smem_bytes;
    // Wavefronts per SIMD the register allocation is held to = workgroups per CU (a workgroup is one wavefront on each SIMD):
    // 4, i.e. 128 VGPRs -- every instantiation fits without scratch after round 3's register diet (the paired 3D kernel: 162 ->
    // 134 VGPRs unconstrained, 128 with no spill), and the LDS admits 4 x 37.5 KB.  The 1D and the unpaired 3D instantiation
    // get there by re-deriving their per-lane LDS addresses every iteration (rederive_lane_addresses: ~30 VALU instructions
    // instead of a dozen loop-invariant VGPRs); the paired 3D and the 2D one fit as they are.  A kernel that spills is worse off
    // than one at 3 wavefronts per SIMD: a scratch reload is a vector-memory load and waits for every prefetch load before it.
    static constexpr bool rederive_lane_addresses = !(Dims == 2 || Paired);
    static constexpr int min_waves_per_simd = 4;
};

// Paired: the two hypercubes of every tile are neighbours along x (3D, 32-bit, even hypercube count along x, aligned
// rows): the tile is fetched as 256 rows of 128 bytes (load_pair_regs) instead of 2 x 256 rows of 64 bytes.
template<typename T, int Dims, bool Aligned, bool Paired = false>
__global__ void __launch_bounds__((tile_cfg<T, Dims>::threads), (db_cfg<T, Dims, Paired>::min_waves_per_simd))
compress_kernel_db(const typename word_of<T>::type *__restrict__ in, const grid_geom gg, uint32_t *__restrict__ header,
        typename word_of<T>::type *__restrict__ body, tile_desc *desc_base, uint32_t *tickets, const uint32_t num_classes,
        uint32_t *out_len, uint32_t len_extra, uint32_t *err) {
    const uint32_t epoch_loaded = launch_epoch_load(tickets);
    using C = tile_cfg<T, Dims>;
    using W = typename C::W;
    using L = typename C::L;
    using P = profile<T, Dims>;
    static_assert(P::B == 32, "register-buffered variant is for 32-bit words");
    constexpr int early_vectors = 4;  // of 8; measured on 512^3: 2 -> 0.221 ms, 4 -> 0.211, 6 -> 0.219 (spills)
    constexpr int K = C::K;
    constexpr int NW = C::threads / 64;
    extern __shared__ __attribute__((aligned(128))) char smem[];

    const int tid = static_cast<int>(threadIdx.x);
    const int grp0 = wave_uniform(tid >> 6) / (threads_per_hc / 64), t = tid % threads_per_hc;
    char *cube = smem + grp0 * C::cube_stride;                            // staging of this group's hypercube
    uint32_t *tile_run = reinterpret_cast<uint32_t *>(smem);             // later: the K encoded runs, back to back
    char *zero_region = smem + K * C::cube_stride;
    // (the second hypercube's staging region sits 64 bytes = 4 sixteen-byte slots further round the banks than the first: so does
    // the zero block its out-of-cube neighbour reads go to -- the whole region is zero)
    char *zero = zero_region + L::template zero_offset<Dims>() + grp0 * (C::cube_stride % 256);
    uint32_t *misc = reinterpret_cast<uint32_t *>(zero_region + L::zero_bytes);  // [0..NW) wave totals, [NW] prefix, [NW+1] next ticket, [NW+2] first ticket

    const uint32_t ntiles = (gg.nhc + K - 1) / K;
    const uint32_t cls = blockIdx.x % num_classes;
    uint32_t *ticket_counter = tickets + cls * ticket_stride_words;
    // Tickets.  The ticket of the tile an iteration works on was drawn a little more than one iteration earlier: the one for
    // the first tile here (its own LDS slot), the one for the second right behind it, every later one in the iteration before,
    // next to the aggregate's publish (behind B2) -- its round trip (~1 us under a streaming load, MI355X guide "dequeue") is
    // over by B3, and its result is looked at only behind the copy-out and the transposes.  Drawn at the top of the iteration
    // that consumes it behind B1 -- as round 1 had it -- that round trip sat on wavefront 0's path to B1 and with it on the
    // whole workgroup's (built without LLVM's atomic optimizer: it turns the single-lane atomicAdd into mbcnt + atomic +
    // readfirstlane and waits for the result at once); drawn behind B3 (rounds 2-4) it sat, through the one in-order memory
    // counter, in the first wait behind the copy-out's stores.  A whole extra tile of look-ahead was measured in round 1 and
    // loses (0.255 vs 0.205 ms: claim order and processing order drift apart and the look-back waits); this is a third of an
    // iteration.
    // (the first ticket's round trip is the first thing on the kernel's critical path: drawn before anything else, with the
    // zero block filled while it is in flight)
    uint32_t first_ticket = 0;
    if (tid == 0) first_ticket = atomicAdd(ticket_counter, 1u);
    for (uint32_t i = tid; i < L::zero_bytes / 4; i += C::threads) reinterpret_cast<uint32_t *>(zero_region)[i] = 0;
    if (tid == 0) misc[NW + 2] = first_ticket;
    __syncthreads();
    const desc_ref desc{desc_base, launch_epoch(epoch_loaded)};
    // (tickets are the same in every lane: as scalars, so that the tile's origin -- two magic-number divisions and 64-bit
    // multiply-adds -- is computed once on the scalar unit, not per lane)
    uint32_t tile = tile_of_ticket(static_cast<uint32_t>(wave_uniform(static_cast<int>(misc[NW + 2]))), cls, num_classes);

    static_assert(!Paired || (Dims == 3 && sizeof(W) == 4 && K == 2 && Aligned), "paired loads: 3D, 32-bit, aligned rows");
    input_regs<W, Aligned> pre;
    if constexpr (Paired) {
        const uint32_t first_tile = tile < ntiles ? tile : ntiles - 1;
        load_pair_regs<>(in, gg, hc_origin<Dims>(gg, first_tile * K), tid, pre);
    } else {
        uint32_t first_hc = tile * K + grp0;
        if (first_hc >= gg.nhc) first_hc = gg.nhc - 1;
        load_hypercube_regs<T, Dims, Aligned>(in, gg, hc_origin<Dims>(gg, first_hc), t, pre);
    }
    if (tid == 0) misc[NW + 1] = atomicAdd(ticket_counter, 1u);  // second ticket, behind the first tile's loads; read behind the first B1
    // the previous tile: transposed planes in registers, aggregate published, waiting for its prefix
    bool have_prev = false, prev_active = false;
    uint32_t prev_tile = 0, prev_aggregate = 0, prev_run_start = 0, prev_my_len = 0, prev_chunk_excl = 0, prev_hc = 0;
    uint32_t prev_head = 0;
    uint32_t planes[32];
#pragma unroll
    for (int j = 0; j < 32; ++j) planes[j] = 0;
    // The loop runs while the workgroup has a tile to ENCODE; the tile it encoded last is written out behind the loop (the drain
    // used to be one more trip through the loop with everything but the write-out switched off: three barriers, the origin
    // arithmetic and eight clamped prefetch loads of a tile nobody needed, per workgroup and launch -- and an `if (have_cur)`
    // around every phase of every iteration).
    while (tile < ntiles) {
        // wave index, lane and the single-lane predicates of this iteration come from a fresh copy of the work-item id
        // (gfx950_lds.hpp: fresh_copy): re-derived here, not carried round the loop as lane masks; addresses keep using `tid` / `t`
        const int tid_i = fresh_copy(tid);
        const int lane = tid_i & 63, wave = wave_uniform(tid_i >> 6), grp = wave / (threads_per_hc / 64);
        const bool first_of_tile = tid_i == 0, first_of_hc = (tid_i & (threads_per_hc - 1)) == 0;
        const int t_i = db_cfg<T, Dims, Paired>::rederive_lane_addresses ? (tid_i & (threads_per_hc - 1)) : t;
        const uint32_t hc = tile * K + grp;
        const bool active = hc < gg.nhc;
        if constexpr (Paired) {
            stage_pair_regs(pre, smem, C::cube_stride, tid);  // (an even hypercube count: both cubes of a tile exist)
        } else {
            if (active) stage_hypercube_regs<W, Aligned>(pre, cube, t_i);
        }
        // The previous tile's look-back window is read BEHIND the staging, not in front of it: the staging's wait for the
        // last prefetched vector is an s_waitcnt vmcnt(0) (gfx9 counts loads and stores in one in-order counter, and the
        // previous copy-out issued a data-dependent number of stores after the prefetch), so a window load issued before it
        // is waited for in full, every iteration, by the wavefront the whole workgroup then waits for at B1 (round 1 had
        // it there: ISA + the ~3 k cycles of its "top of the iteration" phase timer).  Issued here it is in flight until the
        // resolve, most of an iteration later.
        lookback_windows window{};
        if (have_prev && wave == 0) lookback_issue(desc, prev_tile, lane, window);
        __syncthreads();  // B1: cube staged (the next ticket has been in misc[NW + 1] since before the last B4)
        const uint32_t next_tile = tile_of_ticket(static_cast<uint32_t>(wave_uniform(static_cast<int>(misc[NW + 1]))), cls, num_classes);
        __builtin_amdgcn_sched_barrier(0);
        uint32_t next_hc = Paired ? next_tile * K : next_tile * K + grp;
        if (next_hc >= gg.nhc) next_hc = Paired ? gg.nhc - K : gg.nhc - 1;
        const uint64_t next_origin = hc_origin<Dims>(gg, next_hc);
        // early part of the next tile's prefetch (a whole iteration ahead of its use)
        if constexpr (Paired) {
            load_pair_regs<0, early_vectors>(in, gg, next_origin, tid, pre);
        } else {
            load_hypercube_regs<T, Dims, Aligned, 0, early_vectors>(in, gg, next_origin, t, pre);
        }
        __builtin_amdgcn_sched_barrier(0);
        W r[vals_per_thread];
        stencil_residuals<T, Dims>(cube, zero, t_i, r);
        const uint32_t head = chunk_head32(r);
        const uint32_t count = active ? static_cast<uint32_t>(__builtin_popcount(head)) : 0u;
        const uint32_t incl = wave_inclusive_scan(count, lane);
        if (lane == 63) misc[wave] = incl;
        __syncthreads();  // B2: all stencil reads done (staging region reusable), wave totals known
        uint32_t run_start = 0, aggregate = 0, my_len = 0;
#pragma unroll
        for (int g = 0; g < K; ++g) {
            // (wave totals as scalars: the tile's lengths, and with them the copy-out's case analysis, are scalar code, and
            // the previous tile's aggregate / run start / length are carried round the loop in SGPRs, not VGPRs)
            const uint32_t len_g = tile * K + g < gg.nhc ? P::head_words + static_cast<uint32_t>(wave_uniform(static_cast<int>(misc[2 * g])))
                            + static_cast<uint32_t>(wave_uniform(static_cast<int>(misc[2 * g + 1]))) : 0u;
            if (g < grp) run_start += len_g;
            if (g == grp) my_len = len_g;
            aggregate += len_g;
        }
        const uint32_t chunk_excl = ((wave & 1) ? static_cast<uint32_t>(wave_uniform(static_cast<int>(misc[2 * grp]))) : 0u) + incl - count;
        if (first_of_tile) publish_aggregate(desc, tile, aggregate);  // as early as possible: successors wait on this
        // The ticket the NEXT iteration reads behind its B1 (it needs one iff it has a tile), drawn here, by the lane that has just
        // published: its round trip (~1 us under a streaming load) is over long before B3.  Rounds 1-4 drew it right behind B3 so
        // that the copy-out would hide it -- but loads, stores and atomics retire through ONE in-order counter, and the first
        // wait behind the copy-out (whatever it is for) then also waits for this atomic: in the executed stream of the built kernel
        // wavefront 0 stood still for the rest of the round trip in front of its transposes, with the other three waiting at B4.
        const bool draw = first_of_tile && next_tile < ntiles;
        uint32_t ticket_after_next = 0;
        if (draw) ticket_after_next = atomicAdd(ticket_counter, 1u);
        // late part of the prefetch: after the stencil, so these registers are not live across it (the previous
        // tile's planes are).  Both parts are unconditional (clamped index): a conditional load keeps the old registers
        // live around the whole loop.
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (Paired) {
            load_pair_regs<1, early_vectors>(in, gg, next_origin, tid, pre);
        } else {
            load_hypercube_regs<T, Dims, Aligned, 1, early_vectors>(in, gg, next_origin, t, pre);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (have_prev) {
            // the previous tile's planes leave the registers: compact them into the (now free) staging region
            if (prev_active) {
                write_planes32(tile_run + prev_run_start, prev_run_start, t_i, prev_head, prev_chunk_excl, planes);
            }
        }
        // bit-plane transpose of the current tile, in registers; it stays there until the next iteration
#pragma unroll
        for (int j = 0; j < 32; ++j) planes[j] = r[j];
        transpose32(planes);
        __builtin_amdgcn_sched_barrier(0);
        // The previous tile's prefix, as the LAST thing wavefront 0 does before B3: its predecessors (which may lag by
        // a good part of an iteration) have had the most time to publish, and its own late prefetch, which hipcc's
        // vmcnt(0) in the general loop also sits out (loads return in order), has been in flight the longest.  The
        // window read at the top of the iteration settles the prefix without any memory operation in 1-2 of 10 cases.
        // Measured alternatives (512^3 f32, ms): resolve before the late prefetch 0.211; look-back wavefront without
        // prefetch in flight 0.219; window re-read behind B2 and consumed after the plane writes 0.217-0.224; whole
        // resolve right behind B1 0.238 (0.3 polls per tile, convoys); resolve between late prefetch and plane writes
        // 0.207; this order 0.201-0.205.
        if (have_prev && wave == 0) {
            const uint32_t exclusive = resolve_exclusive_prefix_impl<true>(desc, prev_tile, prev_aggregate, err, lane, window);
            if (first_of_tile) misc[NW] = exclusive;
        }
        lds_append_complete();  // (the plane writes of write_planes32 are inline asm: the compiler's waitcnt insertion does not see them)
        __syncthreads();  // B3: previous tile's runs complete in LDS, its prefix known
        if (have_prev) {
            const uint32_t prefix = static_cast<uint32_t>(wave_uniform(static_cast<int>(misc[NW])));
            copy_out<W, C::threads>(reinterpret_cast<const W *>(tile_run), body + prefix, prev_aggregate, tid);
            if (prev_active && first_of_hc) header[prev_hc] = prefix + prev_run_start + prev_my_len;  // offset_after(hc), common.hh:342-347
        }
        // The ticket drawn behind B3 is looked at only HERE, behind the transposes.  hipcc sinks the transposes of the current tile
        // (register-only work, needed next iteration) behind the copy-out on its own -- and, left alone, hoists this conditional
        // LDS store and the s_waitcnt vmcnt(0) in front of its atomic result above them, where a transpose temporary that shares
        // the result's VGPR then makes EVERY wavefront wait for all of its outstanding memory operations: the copy-out's stores
        // just issued and the late prefetch, with ~270 instructions of independent work right behind the wait (round 5, from the
        // executed stream of the built kernel).  With the pin the transposes run while the stores are acknowledged and the
        // atomic returns; the wait moves behind them.
        registers_complete_here(planes);
        if (draw) misc[NW + 1] = ticket_after_next;
        __syncthreads();  // B4: copy-out has read the runs before the next tile is staged over them; next ticket in LDS
        have_prev = true;
        prev_tile = tile;
        prev_aggregate = aggregate;
        prev_run_start = run_start;
        prev_my_len = my_len;
        prev_chunk_excl = chunk_excl;
        prev_head = head;
        prev_active = active;
        prev_hc = hc;
        tile = next_tile;
    }
    // Drain: the tile encoded last is still in registers.  Plane writes, look-back (a window read now: its predecessors have had
    // the whole last iteration), one barrier, copy-out.  A workgroup that never drew a tile (a grid larger than the tile count
    // cannot happen -- launch_persistent clamps it -- but a ticket can lose the race for the last tiles) has nothing to do.
    if (have_prev) {
        const int tid_i = fresh_copy(tid);
        const int lane = tid_i & 63, wave = wave_uniform(tid_i >> 6);
        const bool first_of_tile = tid_i == 0, first_of_hc = (tid_i & (threads_per_hc - 1)) == 0;
        const int t_i = db_cfg<T, Dims, Paired>::rederive_lane_addresses ? (tid_i & (threads_per_hc - 1)) : t;
        if (prev_active) write_planes32(tile_run + prev_run_start, prev_run_start, t_i, prev_head, prev_chunk_excl, planes);
        if (wave == 0) {
            const uint32_t exclusive = resolve_exclusive_prefix(desc, prev_tile, prev_aggregate, err, lane);
            if (first_of_tile) misc[NW] = exclusive;
        }
        lds_append_complete();
        __syncthreads();
        const uint32_t prefix = static_cast<uint32_t>(wave_uniform(static_cast<int>(misc[NW])));
        // (tid_i, not tid: nothing derived from the work-item id is kept alive across the loop for the drain's sake)
        copy_out<W, C::threads>(reinterpret_cast<const W *>(tile_run), body + prefix, prev_aggregate, tid_i);
        if (prev_active && first_of_hc) header[prev_hc] = prefix + prev_run_start + prev_my_len;  // offset_after(hc), common.hh:342-347
        // the last tile ends the body (store_stream_length, cuda_codec.inl:507-511): tiles are drawn in increasing order within a
        // class, so the last tile is always some workgroup's LAST tile -- the length is stored here and nowhere else
        if (first_of_tile && prev_tile == ntiles - 1) store_stream_length(out_len, len_extra + prefix + prev_aggregate);
    }
    // The last workgroup to leave zeroes the ticket counters for the next launch on this handle (stream order makes it
    // visible); the descriptors need no clearing (epoch).
    release_tickets<C::threads>(tickets, num_classes, tid, err, out_len, desc);
}

// ---- the register-buffered deferred-write-out pipeline with 256 work-items per hypercube ("wide" mapping) ---------
// Same iteration structure as compress_kernel_db, one hypercube per tile: stage -> B1 -> early prefetch -> stencil -> B2 ->
// publish -> late prefetch -> plane writes of the previous tile -> transpose of the current one -> look-back (wavefront 0)
// -> B3 -> copy-out -> B4.  See codec_kernels_wide.hpp for the work-item mapping.
template<typename W, int Dims>
struct wide_cfg {
    static constexpr int threads = wide::threads;
    static constexpr int NW = threads / 64;
    static constexpr uint32_t smem_bytes = wide::layout<W>::cube_bytes + wide::layout<W>::zero_bytes + 64;
    // 4 workgroups per CU (128 VGPRs, 4 x 36.4 KB of LDS): every instantiation fits without scratch (see db_cfg)
    static constexpr int min_waves_per_simd = 4;
    static constexpr int early_vectors = wide::input_regs<W>::NV / 2;
};

template<typename W, int Dims, bool Aligned>
__global__ void __launch_bounds__((wide_cfg<W, Dims>::threads), (wide_cfg<W, Dims>::min_waves_per_simd))
compress_kernel_wide(const W *__restrict__ in, const grid_geom gg, uint32_t *__restrict__ header, W *__restrict__ body,
        tile_desc *desc_base, uint32_t *tickets, const uint32_t num_classes, uint32_t *out_len, uint32_t len_extra, uint32_t *err) {
    const uint32_t epoch_loaded = launch_epoch_load(tickets);
    using C = wide_cfg<W, Dims>;
    using L = wide::layout<W>;
    using E = wide::coding<W>;
    constexpr int NW = C::NW;
    constexpr int early_vectors = C::early_vectors;
    extern __shared__ __attribute__((aligned(128))) char smem[];

    const int tid = static_cast<int>(threadIdx.x), t = tid;
    char *cube = smem;
    uint32_t *run32 = reinterpret_cast<uint32_t *>(smem);  // later: the encoded run (f64: as uint32 halves of its words)
    char *zero_region = smem + L::cube_bytes;
    char *zero = zero_region + L::zero_offset;
    uint32_t *misc = reinterpret_cast<uint32_t *>(zero_region + L::zero_bytes);  // [0..NW) wave totals, [NW] prefix, [NW+1] next ticket, [NW+2] first ticket

    const uint32_t ntiles = gg.nhc;
    const uint32_t cls = blockIdx.x % num_classes;
    uint32_t *ticket_counter = tickets + cls * ticket_stride_words;
    // tickets: see compress_kernel_db (first one in its own slot -- drawn first of all, the zero block is filled while it is in
    // flight --, every later one drawn next to the publish of the iteration before its consumer's B1)
    uint32_t first_ticket = 0;
    if (tid == 0) first_ticket = atomicAdd(ticket_counter, 1u);
    for (uint32_t i = tid; i < L::zero_bytes / 4; i += C::threads) reinterpret_cast<uint32_t *>(zero_region)[i] = 0;
    if (tid == 0) misc[NW + 2] = first_ticket;
    __syncthreads();
    const desc_ref desc{desc_base, launch_epoch(epoch_loaded)};
    uint32_t tile = tile_of_ticket(static_cast<uint32_t>(wave_uniform(static_cast<int>(misc[NW + 2]))), cls, num_classes);

    wide::input_regs<W> pre;
    wide::load_regs<W, Dims, Aligned>(in, gg, hc_origin<Dims>(gg, tile < ntiles ? tile : ntiles - 1), t, pre);
    if (tid == 0) misc[NW + 1] = atomicAdd(ticket_counter, 1u);  // second ticket, behind the first tile's loads

    // the previous tile: this lane's plane words in registers, aggregate published, waiting for its prefix
    bool have_prev = false;
    uint32_t prev_tile = 0, prev_aggregate = 0;
    typename E::held prev_held{};
    uint32_t planes[E::planes_per_lane];
#pragma unroll
    for (int j = 0; j < E::planes_per_lane; ++j) planes[j] = 0;
    // (the loop runs while there is a tile to encode; the last one is written out behind it: see compress_kernel_db)
    while (tile < ntiles) {
        // (wave index, lane and the single-lane predicate re-derived per iteration: see compress_kernel_db.  Here the LDS
        // addresses of the staging, the stencil's row pointers and the coding's lane roles are re-derived from it too: ~30 VALU
        // instructions per iteration instead of a dozen loop-invariant VGPRs, which is what fits the 1D and 2D kernel into 128)
        const int tid_i = fresh_copy(tid);
        const int lane = tid_i & 63, wave = wave_uniform(tid_i >> 6);
        const bool first_of_tile = tid_i == 0;
        wide::stage_regs<W>(pre, cube, tid_i);
        lookback_windows window{};  // (behind the staging: see compress_kernel_db)
        if (have_prev && wave == 0) lookback_issue(desc, prev_tile, lane, window);
        __syncthreads();  // B1: cube staged (the next ticket has been in misc[NW + 1] since before the last B4)
        const uint32_t next_tile = tile_of_ticket(static_cast<uint32_t>(wave_uniform(static_cast<int>(misc[NW + 1]))), cls, num_classes);
        __builtin_amdgcn_sched_barrier(0);
        const uint64_t next_origin = hc_origin<Dims>(gg, next_tile < ntiles ? next_tile : ntiles - 1);
        wide::load_regs<W, Dims, Aligned, 0, early_vectors>(in, gg, next_origin, t, pre);
        __builtin_amdgcn_sched_barrier(0);
        W r[wide::vals];
        uint32_t head_a = 0, head_b = 0;
        wide::stencil<W, Dims>(cube, zero, tid_i, r);
        const uint32_t count = E::head_and_count(r, head_a, head_b);
        // (the lanes of a chunk end up with the same inclusive value: only the first one feeds the scan)
        const uint32_t incl = wave_inclusive_scan((tid_i & (E::lanes_per_chunk - 1)) == 0 ? count : 0u, lane);
        if (lane == 63) misc[wave] = incl;
        __syncthreads();  // B2: all stencil reads done (staging region reusable), wave totals known
        uint32_t aggregate = E::head_words, chunk_excl = 0;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            const uint32_t total = static_cast<uint32_t>(wave_uniform(static_cast<int>(misc[w])));  // (scalars: see compress_kernel_db)
            aggregate += total;
            if (w < wave) chunk_excl += total;
        }
        chunk_excl += incl - count;
        if (first_of_tile) publish_aggregate(desc, tile, aggregate);  // as early as possible: successors wait on this
        const bool draw = first_of_tile && next_tile < ntiles;  // the ticket the next iteration reads behind its B1: see compress_kernel_db
        uint32_t ticket_after_next = 0;
        if (draw) ticket_after_next = atomicAdd(ticket_counter, 1u);
        __builtin_amdgcn_sched_barrier(0);
        wide::load_regs<W, Dims, Aligned, 1, early_vectors>(in, gg, next_origin, t, pre);
        __builtin_amdgcn_sched_barrier(0);
        // the previous tile's planes leave the registers: compact them into the (now free) staging region
        if (have_prev) E::write(prev_held, planes, run32, tid_i);
        // bit-plane transpose of the current tile, in registers; it stays there until the next iteration
        E::transpose(r, tid_i, planes);
        __builtin_amdgcn_sched_barrier(0);
        if (have_prev && wave == 0) {
            const uint32_t exclusive = resolve_exclusive_prefix_impl<true>(desc, prev_tile, prev_aggregate, err, lane, window);
            if (first_of_tile) misc[NW] = exclusive;
        }
        lds_append_complete();  // (E::write's plane stores are inline asm: the compiler's waitcnt insertion does not see them)
        __syncthreads();  // B3: previous tile's run complete in LDS, its prefix known
        if (have_prev) {
            const uint32_t prefix = static_cast<uint32_t>(wave_uniform(static_cast<int>(misc[NW])));
            copy_out<W, C::threads>(reinterpret_cast<const W *>(smem), body + prefix, prev_aggregate, tid);
            if (first_of_tile) header[prev_tile] = prefix + prev_aggregate;  // offset_after(hc), common.hh:342-347
        }
        // (the ticket is looked at behind the transposes, which hipcc sinks behind the copy-out: see compress_kernel_db)
        static_assert(E::planes_per_lane == 32, "registers_complete_here takes the 32 plane registers");
        registers_complete_here(planes);
        if (draw) misc[NW + 1] = ticket_after_next;
        __syncthreads();  // B4: copy-out has read the run before the next tile is staged over it; next ticket in LDS
        have_prev = true;
        prev_tile = tile;
        prev_aggregate = aggregate;
        prev_held = E::hold(tid_i, head_a, head_b, E::head_words + chunk_excl);
        tile = next_tile;
    }
    // drain: the tile encoded last is still in registers (see compress_kernel_db)
    if (have_prev) {
        const int tid_i = fresh_copy(tid);
        const int lane = tid_i & 63, wave = wave_uniform(tid_i >> 6);
        const bool first_of_tile = tid_i == 0;
        E::write(prev_held, planes, run32, tid_i);
        if (wave == 0) {
            const uint32_t exclusive = resolve_exclusive_prefix(desc, prev_tile, prev_aggregate, err, lane);
            if (first_of_tile) misc[NW] = exclusive;
        }
        lds_append_complete();
        __syncthreads();
        const uint32_t prefix = static_cast<uint32_t>(wave_uniform(static_cast<int>(misc[NW])));
        copy_out<W, C::threads>(reinterpret_cast<const W *>(smem), body + prefix, prev_aggregate, tid_i);
        if (first_of_tile) {
            header[prev_tile] = prefix + prev_aggregate;  // offset_after(hc), common.hh:342-347
            // (the last hypercube is always some workgroup's last tile: the stream length is stored here and nowhere else)
            if (prev_tile == gg.nhc - 1) {
                store_stream_length(out_len, len_extra + prefix + prev_aggregate);
                // zero the header pad of 64-bit streams with an odd hypercube count (cuda_codec.inl:446-452)
                if (sizeof(W) == 8 && (gg.nhc & 1u)) header[gg.nhc] = 0;
            }
        }
    }
    release_tickets<C::threads>(tickets, num_classes, tid, err, out_len, desc);
}

template<typename T, int Dims, bool Aligned>
__global__ void __launch_bounds__((tile_cfg<T, Dims>::threads))
decompress_kernel(const uint32_t *__restrict__ header, const uint32_t *__restrict__ header_base_ptr, const typename word_of<T>::type *__restrict__ body,
        typename word_of<T>::type *__restrict__ out, const grid_geom gg, uint32_t *err, const uint32_t body_words, const uint32_t num_xcds) {
    using C = tile_cfg<T, Dims>;
    using W = typename C::W;
    using L = typename C::L;
    using P = profile<T, Dims>;
    constexpr int K = C::K;
    extern __shared__ __attribute__((aligned(128))) char smem[];

    const int tid = static_cast<int>(threadIdx.x);
    const int grp = wave_uniform(tid / threads_per_hc), t = tid % threads_per_hc;
    char *cube = smem + grp * C::cube_stride;
    uint32_t *xchg = reinterpret_cast<uint32_t *>(smem + K * C::cube_stride + L::zero_bytes) + grp * (C::xchg_bytes / 4);

    // Workgroups are dealt to the XCDs round-robin (block b runs on XCD b % num_xcds; num_xcds = hipDeviceAttributeNumberOfXccs
    // of the device the launch goes to: 8 in SPX mode, fewer in a partitioned device), each with its own L2.  Giving every XCD a
    // CONTIGUOUS range of tiles makes neighbouring tiles -- whose hypercube rows share cache lines when the rows are not
    // line-aligned, and whose streams are adjacent -- meet in one L2 instead of leaving two partially written copies of a
    // line in two L2s (510x511x509 f32: 0.313 -> 0.239 ms; aligned grids unchanged).  The mapping is a permutation of the
    // tiles whatever num_xcds is: a wrong count costs locality, never correctness.
    // For f64 with aligned rows (every hypercube row is whole lines) the plain order measured 2-4 % faster: used there.
    constexpr bool xcd_ranges = sizeof(W) == 4 || !Aligned;
    const uint32_t ntiles = (gg.nhc + K - 1) / K;
    const uint32_t per_xcd = (ntiles + num_xcds - 1) / num_xcds;
    const uint32_t tile = xcd_ranges ? (blockIdx.x % num_xcds) * per_xcd + blockIdx.x / num_xcds : blockIdx.x;
    const uint32_t hc = tile < ntiles ? tile * K + grp : gg.nhc;  // (grid rounded up to a multiple of num_xcds: surplus blocks idle)
    const bool active = hc < gg.nhc;
    uint32_t begin = 0, len = 0;
    if (active) {
        const uint32_t header_base = header_base_ptr ? *header_base_ptr : 0u;
        begin = (hc ? header[hc - 1] : header_base) - header_base;  // stream<Profile>::hypercube, common.hh:350-358
        len = header[hc] - header_base - begin;
        // A header entry is trusted only as far as the format allows: every hypercube takes head_words..max_hc_words
        // words, so offset_after(hc - 1) lies in [hc * head_words, hc * max_hc_words], and the run must end inside the
        // body the caller vouched for (body_words; 0xffffffff = unknown, then the format bound is all there is).  A
        // corrupt entry makes this hypercube decode as zeros and sets the error word instead of reading wherever
        // `body + begin` points.
        const uint64_t lo = static_cast<uint64_t>(hc) * P::head_words, hi = static_cast<uint64_t>(hc) * P::max_hc_words;
        if (len < static_cast<uint32_t>(P::head_words) || len > static_cast<uint32_t>(P::max_hc_words) || begin < lo || begin > hi
                || static_cast<uint64_t>(begin) + len > body_words) {
            if (t == 0) atomicOr(err, 2u);  // corrupt header
            len = 0;
        }
    }
    // Load the encoded run with 16-byte vectors: start at the 16-byte boundary at or below the run's first word
    // (never leaves the aligned block that holds valid stream words), so the run sits `mis` words into the LDS
    // region.  All loads of a work-item are issued before the first one is consumed.
    constexpr uint32_t wpv = 16 / sizeof(W);
    constexpr int max_vec = (P::max_hc_words + wpv - 1 + wpv - 1) / wpv;            // run + worst misalignment
    constexpr int vec_per_thread = (max_vec + threads_per_hc - 1) / threads_per_hc;   // 9 (f32) / 17 (f64)
    uint32_t mis = 0;
    if (len == 0) {
        // padding group or corrupt entry: decode an all-zero hypercube so every LDS index stays in range
        if (t < P::head_words) *run_layout<W>::ptr(reinterpret_cast<W *>(cube) + t) = 0;
    } else {
        const W *src = body + begin;
        mis = static_cast<uint32_t>((reinterpret_cast<uintptr_t>(src) / sizeof(W)) % wpv);
        const vec16 *src16 = reinterpret_cast<const vec16 *>(src - mis);
        const uint32_t nvec = (len + mis + wpv - 1) / wpv;
        // 32-bit profiles, branch-free: every work-item loads and stores all of its vector slots.  Slots behind the run
        // (j >= nvec: about a third of them on the benchmark data) load the run's LAST block again (a valid address; the lanes of
        // a wave-instruction that do so hit one line) and park the copy behind the run in LDS, where nothing looks -- the staging
        // region holds exactly vec_per_thread * 128 vectors.  (The conditional form is 18 exec-mask diamonds in a kernel of ~330
        // scalar-side instructions: 327 -> 269, and 42 VALU instructions less.)  64-bit profiles keep the conditional form: their
        // runs fill four fifths of the slots and every parked vector would pay the run layout's swizzle arithmetic.
        static_assert(static_cast<uint32_t>(vec_per_thread) * threads_per_hc * 16u <= L::cube_bytes, "parked vectors stay inside the staging region");
        constexpr bool branch_free = sizeof(W) == 4;
        vec16 v[vec_per_thread];
#pragma unroll
        for (int i = 0; i < vec_per_thread; ++i) {
            const uint32_t j = static_cast<uint32_t>(i * threads_per_hc + t);
            if constexpr (branch_free) {
                // (wave-uniform base: the run's first block; per lane: 16 bytes x the slot, 32 bits)
                v[i] = global_load16_block(reinterpret_cast<const char *>(scalar_pointer(src16)) + lane_offset_here(16u * (j < nvec ? j : nvec - 1)));
            } else {
                if (j < nvec) v[i] = global_load16_block(reinterpret_cast<const char *>(scalar_pointer(src16)) + lane_offset_here(16u * j));  // (read once; the end blocks may hold foreign words)
            }
        }
#pragma unroll
        for (int i = 0; i < vec_per_thread; ++i) {
            const uint32_t j = static_cast<uint32_t>(i * threads_per_hc + t);
            if (branch_free || j < nvec) lds_write16(run_layout<W>::ptr(cube + 16 * j), v[i]);
        }
    }
    __syncthreads();
    // (a hypercube whose header entry was rejected decodes the all-zero run staged above: its region of `out` is written as
    // zeros -- deterministic, never what the caller's buffer happened to hold -- and the error word says so)
    decode_hypercube<T, Dims, Aligned>(out, gg, active ? hc_origin<Dims>(gg, hc) : 0, active, cube,
            mis * static_cast<uint32_t>(sizeof(W)), xchg, t);
}

// ---- f64 with 256 work-items per hypercube (codec_kernels_wide.hpp: wide::decode_residuals / wide::inverse_transform) --------
// Same header lookup, bounds and run staging as decompress_kernel; one hypercube per workgroup.  4 workgroups per CU by LDS
// (35-36 KB each) = 16 wavefronts = 4 per SIMD, which the registers are held to (the 128-work-item kernel: 8 wavefronts per CU).
template<int Dims>
struct wide_decode_cfg {
    static constexpr int threads = wide::threads;
    static constexpr uint32_t smem_bytes = wide::decode_layout<Dims>::smem_bytes;
    static constexpr int min_waves_per_simd = 4;
};

template<int Dims, bool Aligned>
__global__ void __launch_bounds__((wide_decode_cfg<Dims>::threads), (wide_decode_cfg<Dims>::min_waves_per_simd))
decompress_kernel_wide(const uint32_t *__restrict__ header, const uint32_t *__restrict__ header_base_ptr, const uint64_t *__restrict__ body,
        uint64_t *__restrict__ out, const grid_geom gg, uint32_t *err, const uint32_t body_words, const uint32_t num_xcds) {
    using W = uint64_t;
    using P = profile<double, Dims>;
    using D = wide::decode_layout<Dims>;
    constexpr int threads = wide::threads;
    extern __shared__ __attribute__((aligned(128))) char smem[];
    const int t = static_cast<int>(threadIdx.x);
    char *cube = smem;
    uint32_t *totals = reinterpret_cast<uint32_t *>(smem + D::totals_offset);

    // (tile order: see decompress_kernel -- aligned f64 rows are whole cache lines and keep the plain order)
    constexpr bool xcd_ranges = !Aligned;
    const uint32_t ntiles = gg.nhc;
    const uint32_t per_xcd = (ntiles + num_xcds - 1) / num_xcds;
    const uint32_t hc = xcd_ranges ? (blockIdx.x % num_xcds) * per_xcd + blockIdx.x / num_xcds : blockIdx.x;
    const bool active = hc < gg.nhc;  // (grid rounded up to a multiple of num_xcds: surplus blocks idle)
    uint32_t begin = 0, len = 0;
    if (active) {
        // header entries are trusted only as far as the format allows: see decompress_kernel
        const uint32_t header_base = header_base_ptr ? *header_base_ptr : 0u;
        begin = (hc ? header[hc - 1] : header_base) - header_base;  // stream<Profile>::hypercube, common.hh:350-358
        len = header[hc] - header_base - begin;
        const uint64_t lo = static_cast<uint64_t>(hc) * P::head_words, hi = static_cast<uint64_t>(hc) * P::max_hc_words;
        if (len < static_cast<uint32_t>(P::head_words) || len > static_cast<uint32_t>(P::max_hc_words) || begin < lo || begin > hi
                || static_cast<uint64_t>(begin) + len > body_words) {
            if (t == 0) atomicOr(err, 2u);  // corrupt header
            len = 0;
        }
    }
    constexpr uint32_t wpv = 16 / sizeof(W);
    constexpr int max_vec = (P::max_hc_words + wpv - 1 + wpv - 1) / wpv;  // run + worst misalignment
    constexpr int vec_per_thread = (max_vec + threads - 1) / threads;      // 9
    static_assert(static_cast<uint32_t>(max_vec) * 16u + 128u <= D::region_bytes, "run, and the plane word read behind it, inside the region");
    uint32_t mis = 0;
    if (len == 0) {
        // padding workgroup or corrupt entry: decode an all-zero hypercube so every LDS index stays in range
        if (t < P::head_words) *run_layout<W>::ptr(reinterpret_cast<W *>(cube) + t) = 0;
    } else {
        const W *src = body + begin;
        mis = static_cast<uint32_t>((reinterpret_cast<uintptr_t>(src) / sizeof(W)) % wpv);
        const vec16 *src16 = reinterpret_cast<const vec16 *>(src - mis);
        const uint32_t nvec = (len + mis + wpv - 1) / wpv;
        vec16 v[vec_per_thread];  // (all loads of a work-item are issued before the first one is consumed)
#pragma unroll
        for (int i = 0; i < vec_per_thread; ++i) {
            const uint32_t j = static_cast<uint32_t>(i * threads + t);
            if (j < nvec) v[i] = global_load16_block(reinterpret_cast<const char *>(scalar_pointer(src16)) + lane_offset_here(16u * j));
        }
#pragma unroll
        for (int i = 0; i < vec_per_thread; ++i) {
            const uint32_t j = static_cast<uint32_t>(i * threads + t);
            if (j < nvec) lds_write16(run_layout<W>::ptr(cube + 16 * j), v[i]);
        }
    }
    __syncthreads();
    W r[wide::vals];
    wide::decode_residuals(cube, mis * static_cast<uint32_t>(sizeof(W)), totals, t, r);
    wide::inverse_transform<Dims, Aligned>(r, out, gg, active ? hc_origin<Dims>(gg, hc) : 0, active, cube, smem, t);
}

#ifdef NDZIP_STAGE_KERNELS
// ---- stage kernels for the parity tests: exactly one hypercube, through the SAME device functions the production kernels
// call (mirror of the reference's stage tests, src/test/codec_profile_test.inl:514-549, :552-729, :735-801, :889-947) --------

// 128 work-items: the f32 encode stages (stage_hypercube_regs / stencil_residuals / chunk_head32 / transpose32 /
// write_planes32 = what compress_kernel_db runs) and the decode stages of both types (decode_residuals /
// inverse_transform_hypercube = what decompress_kernel runs)
template<typename T, int Dims, bool Aligned>
__global__ void __launch_bounds__(threads_per_hc)
debug_stage_kernel(int stage, const grid_geom gg, uint32_t hc, const typename word_of<T>::type *__restrict__ in,
        typename word_of<T>::type *__restrict__ out, uint32_t *out_len) {
    using W = typename word_of<T>::type;
    using L = lds_layout<W>;
    using P = profile<T, Dims>;
    extern __shared__ __attribute__((aligned(128))) char smem[];
    char *cube = smem;
    char *zero_region = smem + L::cube_bytes;
    char *zero = zero_region + L::template zero_offset<Dims>();
    uint32_t *xchg = reinterpret_cast<uint32_t *>(zero_region + L::zero_bytes);
    const int t = static_cast<int>(threadIdx.x);
    const int lane = t & 63, wave = t >> 6;
    for (uint32_t i = t; i < L::zero_bytes / 4; i += threads_per_hc) reinterpret_cast<uint32_t *>(zero_region)[i] = 0;

    W r[vals_per_thread];
    if (stage == debug_forward_transform) {
        forward_transform_hypercube<T, Dims, Aligned>(in, gg, hc_origin<Dims>(gg, hc), true, cube, zero, t, r);
        for (int j = 0; j < vals_per_thread; ++j) out[t * 32 + j] = r[j];
    } else if (stage == debug_encode_residuals) {
        if constexpr (sizeof(W) == 4) {
            for (int j = 0; j < vals_per_thread; ++j) r[j] = in[t * 32 + j];
            const uint32_t head = chunk_head32(r);
            const uint32_t count = static_cast<uint32_t>(__builtin_popcount(head));
            const uint32_t incl = wave_inclusive_scan(count, lane);
            if (lane == 63) xchg[wave] = incl;
            __syncthreads();
            uint32_t planes[32];
#pragma unroll
            for (int j = 0; j < 32; ++j) planes[j] = r[j];
            transpose32(planes);
            write_planes32(reinterpret_cast<uint32_t *>(cube), 0, t, head, (wave ? xchg[0] : 0u) + incl - count, planes);
            lds_append_complete();
            __syncthreads();
            const uint32_t len = P::head_words + xchg[0] + xchg[1];
            const W *src = reinterpret_cast<const W *>(cube);
            for (uint32_t w = t; w < len; w += threads_per_hc) out[w] = src[w];
            if (t == 0) *out_len = len;
        }
    } else if (stage == debug_decode_residuals) {
        for (uint32_t w = t; w < P::max_hc_words; w += threads_per_hc) {
            *run_layout<W>::ptr(reinterpret_cast<W *>(cube) + w) = in[w];
        }
        __syncthreads();
        decode_residuals<T, Dims>(cube, 0, xchg, t, r);
        for (int j = 0; j < vals_per_thread; ++j) out[t * 32 + j] = r[j];
    } else if (stage == debug_inverse_transform) {
        for (int j = 0; j < vals_per_thread; ++j) r[j] = in[t * 32 + j];
        __syncthreads();
        inverse_transform_hypercube<T, Dims, Aligned>(r, out, gg, hc_origin<Dims>(gg, hc), true, cube, xchg, t);
    }
}

// 256 work-items: the f64 encode stages (wide::load_regs / stage_regs / stencil / coding = what compress_kernel_wide runs)
template<int Dims, bool Aligned>
__global__ void __launch_bounds__(wide::threads)
debug_stage_wide_kernel(int stage, const grid_geom gg, uint32_t hc, const uint64_t *__restrict__ in, uint64_t *__restrict__ out,
        uint32_t *out_len) {
    using W = uint64_t;
    using L = wide::layout<W>;
    using E = wide::coding<W>;
    constexpr int NW = wide::threads / 64;
    extern __shared__ __attribute__((aligned(128))) char smem[];
    char *cube = smem;
    char *zero_region = smem + L::cube_bytes;
    char *zero = zero_region + L::zero_offset;
    uint32_t *misc = reinterpret_cast<uint32_t *>(zero_region + L::zero_bytes);
    const int t = static_cast<int>(threadIdx.x);
    const int lane = t & 63, wave = t >> 6;
    for (uint32_t i = t; i < L::zero_bytes / 4; i += wide::threads) reinterpret_cast<uint32_t *>(zero_region)[i] = 0;

    W r[wide::vals];
    if (stage == debug_forward_transform) {
        wide::input_regs<W> pre;
        wide::load_regs<W, Dims, Aligned>(in, gg, hc_origin<Dims>(gg, hc), t, pre);
        wide::stage_regs<W>(pre, cube, t);
        __syncthreads();
        wide::stencil<W, Dims>(cube, zero, t, r);
        for (int j = 0; j < wide::vals; ++j) out[t * wide::vals + j] = r[j];
    } else if (stage == debug_encode_residuals) {
        for (int j = 0; j < wide::vals; ++j) r[j] = in[t * wide::vals + j];
        uint32_t head_a = 0, head_b = 0;
        const uint32_t count = E::head_and_count(r, head_a, head_b);
        const uint32_t incl = wave_inclusive_scan((t & (E::lanes_per_chunk - 1)) == 0 ? count : 0u, lane);
        if (lane == 63) misc[wave] = incl;
        __syncthreads();
        uint32_t total = E::head_words, chunk_excl = 0;
        for (int w = 0; w < NW; ++w) {
            total += misc[w];
            if (w < wave) chunk_excl += misc[w];
        }
        chunk_excl += incl - count;
        uint32_t planes[E::planes_per_lane];
        E::transpose(r, t, planes);
        E::write(E::hold(t, head_a, head_b, E::head_words + chunk_excl), planes, reinterpret_cast<uint32_t *>(cube), t);
        lds_append_complete();
        __syncthreads();
        for (uint32_t w = t; w < total; w += wide::threads) {
            out[w] = *run_layout<W>::ptr(reinterpret_cast<const W *>(cube) + w);
        }
        if (t == 0) *out_len = total;
    } else if (stage == debug_decode_residuals_wide) {
        // (the decode stages use decode_layout: the run / the values in [0, cube_bytes), the exchange words behind them -- the
        // launcher sizes the LDS for both layouts)
        using P = profile<double, Dims>;
        for (uint32_t w = t; w < P::max_hc_words; w += wide::threads) {
            *run_layout<W>::ptr(reinterpret_cast<W *>(cube) + w) = in[w];
        }
        __syncthreads();
        wide::decode_residuals(cube, 0, reinterpret_cast<uint32_t *>(smem + wide::decode_layout<Dims>::totals_offset), t, r);
        // (the production path undoes complement_negative in the plane domain; the stage's contract is the residuals as encoded)
        for (int j = 0; j < wide::vals; ++j) out[t * wide::vals + j] = complement_negative(r[j]);
    } else if (stage == debug_inverse_transform_wide) {
        for (int j = 0; j < wide::vals; ++j) r[j] = complement_negative(in[t * wide::vals + j]);
        __syncthreads();
        wide::inverse_transform<Dims, Aligned>(r, out, gg, hc_origin<Dims>(gg, hc), true, cube, smem, t);
    }
}

__global__ void debug_transpose_kernel(const uint32_t *in, uint32_t *out, uint32_t n, int generic) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t x[32];
#pragma unroll
    for (int j = 0; j < 32; ++j) x[j] = in[i * 32 + j];
    if (generic) {
        transpose32_generic(x);
    } else {
        transpose32(x);
    }
#pragma unroll
    for (int j = 0; j < 32; ++j) out[i * 32 + j] = x[j];
}

// the DPP wave scan and wave sum on their own (one wavefront per 64 inputs)
__global__ void debug_wave_scan_kernel(const uint32_t *in, uint32_t *out, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;  // (the grid covers n exactly: every lane of every wavefront is live)
    const int lane = static_cast<int>(threadIdx.x & 63u);
    const uint32_t v = in[i];
    out[i] = wave_inclusive_scan(v, lane);
    const uint32_t total = wave_sum(v);
    if (lane == 0) out[n + i / 64] = total;
}

// The device-wide scan on its own: the production ticket scheme, aggregate publish, look-back (preloaded window one iteration
// later, exactly the compress kernels' order: publish tile i, then resolve tile i-1) and release, over `ntiles` given lengths
// instead of encoded hypercubes.  One wavefront per workgroup -- the look-back is wavefront 0's job in the compress kernels too.
// (Reference counterpart of what this checks: hierarchical_inclusive_scan at 2^24 elements, src/test/cuda_bits_test.cu:94-114.)
__global__ void __launch_bounds__(64)
debug_lookback_kernel(const uint32_t *__restrict__ lengths, uint32_t *__restrict__ exclusive, uint32_t ntiles, tile_desc *desc_base,
        uint32_t *tickets, const uint32_t num_classes, uint32_t *total, uint32_t *err) {
    const uint32_t epoch_loaded = launch_epoch_load(tickets);
    __shared__ uint32_t slot[2];
    const int lane = static_cast<int>(threadIdx.x);
    const uint32_t cls = blockIdx.x % num_classes;
    uint32_t *ticket_counter = tickets + cls * ticket_stride_words;
    if (lane == 0) slot[0] = atomicAdd(ticket_counter, 1u);
    __syncthreads();
    const desc_ref desc{desc_base, launch_epoch(epoch_loaded)};
    uint32_t tile = tile_of_ticket(static_cast<uint32_t>(wave_uniform(static_cast<int>(slot[0]))), cls, num_classes);
    bool have_prev = false;
    uint32_t prev_tile = 0, prev_aggregate = 0;
    for (;;) {
        const bool have_cur = tile < ntiles;
        if (!have_cur && !have_prev) break;
        lookback_windows window{};
        if (have_prev) lookback_issue(desc, prev_tile, lane, window);
        uint32_t aggregate = 0;
        if (have_cur) {
            aggregate = lengths[tile];
            if (lane == 0) publish_aggregate(desc, tile, aggregate);
        }
        if (have_prev) {
            const uint32_t prefix = resolve_exclusive_prefix_impl<true>(desc, prev_tile, prev_aggregate, err, lane, window);
            if (lane == 0) {
                exclusive[prev_tile] = prefix;
                if (prev_tile == ntiles - 1) store_stream_length(total, prefix + prev_aggregate);
            }
        }
        uint32_t next_tile = tile;
        if (have_cur) {
            if (lane == 0) slot[1] = atomicAdd(ticket_counter, 1u);
            __syncthreads();
            next_tile = tile_of_ticket(static_cast<uint32_t>(wave_uniform(static_cast<int>(slot[1]))), cls, num_classes);
            __syncthreads();
        }
        have_prev = have_cur;
        prev_tile = tile;
        prev_aggregate = aggregate;
        tile = next_tile;
    }
    release_tickets<64>(tickets, num_classes, lane, err, total, desc);
}

#endif  // NDZIP_STAGE_KERNELS

// ---- launchers ----------------------------------------------------------------------------------------------------------

// persistent grid, fully resident: workgroups per CU bounded by the occupancy query and by what the LDS alone admits
template<typename Kernel>
hipError_t persistent_blocks_per_cu(Kernel kernel, int threads, uint32_t smem_bytes, int *out) {
    int api = 0;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
            static_cast<int>(smem_bytes));
    if (e != hipSuccess) return e;
    e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&api, kernel, threads, smem_bytes);
    if (e != hipSuccess) return e;
    const int by_lds = static_cast<int>((160u * 1024u) / smem_bytes);
    int n = api < by_lds ? api : by_lds;
    *out = n < 1 ? 1 : n;
    return hipSuccess;
}

// Workgroups per CU of one kernel, asked of the runtime once PER DEVICE (a process that drives several GPUs launches the same
// kernel on parts with different CU / LDS configurations; ordinals beyond the table are simply asked every time).
struct occupancy_cache {
    static constexpr int max_devices = 64;
    int blocks_per_cu[max_devices] = {};
    template<typename Kernel>
    hipError_t get(int device, Kernel kernel, int threads, uint32_t smem_bytes, int *out) {
        if (device >= 0 && device < max_devices && blocks_per_cu[device] != 0) {
            *out = blocks_per_cu[device];
            return hipSuccess;
        }
        const hipError_t e = persistent_blocks_per_cu(kernel, threads, smem_bytes, out);
        if (e == hipSuccess && device >= 0 && device < max_devices) blocks_per_cu[device] = *out;  // (racing threads store the same value)
        return e;
    }
};

// Scratch layout (fixed, whatever the extent): [16 x u64 reserved (lab builds: phase counters)][ticket counters, one per 128 B][1 line:
// workgroups done][1 line: launch epoch, descriptor count][descriptors].  Nothing is cleared per launch and nothing about a launch
// comes from the host: descriptors carry the launch epoch, which the kernel reads from the scratch and advances on its way out,
// where it also zeroes the ticket counters (the owner of the scratch zeroes everything once and sets epoch 1: init_scratch_epoch).
template<typename Kernel, typename W>
hipError_t launch_persistent(Kernel kernel, int threads, uint32_t smem_bytes, int blocks_per_cu, uint32_t ntiles, const compress_args &a) {
    if (a.max_blocks_per_cu > 0 && blocks_per_cu > a.max_blocks_per_cu) blocks_per_cu = a.max_blocks_per_cu;
    uint32_t grid = static_cast<uint32_t>(a.num_cus) * static_cast<uint32_t>(blocks_per_cu);
    if (grid > ntiles) grid = ntiles;
    // (a grid smaller than the class count would leave classes without a workgroup, i.e. tiles nobody draws)
    const uint32_t num_classes = grid < ticket_classes ? 1u : ticket_classes;
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(threads), smem_bytes, a.stream, static_cast<const W *>(a.in), a.gg, a.header,
            static_cast<W *>(a.body), a.desc + scratch_extra_descs, reinterpret_cast<uint32_t *>(a.desc + 16), num_classes, a.out_len,
            a.len_extra, a.err);
    return hipGetLastError();
}

template<typename T, int Dims, bool Aligned>
hipError_t launch_compress_profile(const compress_args &a) {
    using W = typename word_of<T>::type;
    if (a.gg.nhc == 0) return hipSuccess;
    if constexpr (sizeof(T) == 8) {
        // f64: 256 work-items per hypercube, one hypercube per tile (codec_kernels_wide.hpp)
        using C = wide_cfg<W, Dims>;
        auto kernel = compress_kernel_wide<W, Dims, Aligned>;
        static occupancy_cache cache;
        int blocks_per_cu = 0;
        if (hipError_t e = cache.get(a.device, kernel, C::threads, C::smem_bytes, &blocks_per_cu); e != hipSuccess) return e;
        return launch_persistent<decltype(kernel), W>(kernel, C::threads, C::smem_bytes, blocks_per_cu, a.gg.nhc, a);
    } else {
        // f32: 128 work-items per hypercube, two hypercubes per tile; in 3D with an even hypercube count along x every tile
        // 2m, 2m+1 is a pair of x-neighbours and is fetched as 256 rows of 128 bytes
        using C = tile_cfg<T, Dims>;
        const uint32_t ntiles = (a.gg.nhc + C::K - 1) / C::K;
        bool paired = false;
        auto kernel = compress_kernel_db<T, Dims, Aligned, false>;
        if constexpr (Dims == 3 && Aligned) {
            paired = a.gg.g[2] % 2 == 0;
            if (paired) kernel = compress_kernel_db<T, Dims, Aligned, true>;
        }
        static occupancy_cache cache_of[2];
        int blocks_per_cu = 0;
        if (hipError_t e = cache_of[paired ? 1 : 0].get(a.device, kernel, C::threads, C::smem_bytes, &blocks_per_cu); e != hipSuccess) return e;
        return launch_persistent<decltype(kernel), W>(kernel, C::threads, C::smem_bytes, blocks_per_cu, ntiles, a);
    }
}

template<typename T, int Dims, bool Aligned>
hipError_t launch_decompress_profile(const decompress_args &a) {
    using C = tile_cfg<T, Dims>;
    using W = typename C::W;
    const uint32_t ntiles = (a.gg.nhc + C::K - 1) / C::K;
    if (ntiles == 0) return hipSuccess;
    const uint32_t xcds = a.num_xcds > 0 ? static_cast<uint32_t>(a.num_xcds) : 1u;
    if constexpr (sizeof(T) == 8) {
        // f64: the mapping the caller chose on the handle (an A/B switch), else default_f64_work_items (codec_launch.hpp)
        if ((a.f64_work_items ? a.f64_work_items : default_f64_work_items) == 256) {
            using WC = wide_decode_cfg<Dims>;
            const uint32_t grid = (a.gg.nhc + xcds - 1) / xcds * xcds;
            hipLaunchKernelGGL((decompress_kernel_wide<Dims, Aligned>), dim3(grid), dim3(WC::threads), WC::smem_bytes, a.stream, a.header,
                    a.header_base, static_cast<const W *>(a.body), static_cast<W *>(a.out), a.gg, a.err, a.body_words, xcds);
            return hipGetLastError();
        }
    }
    const uint32_t grid = (ntiles + xcds - 1) / xcds * xcds;  // see the kernel: tiles are dealt to XCDs in contiguous ranges
    hipLaunchKernelGGL((decompress_kernel<T, Dims, Aligned>), dim3(grid), dim3(C::threads), C::smem_bytes, a.stream,
            a.header, a.header_base, static_cast<const W *>(a.body), static_cast<W *>(a.out), a.gg, a.err, a.body_words, xcds);
    return hipGetLastError();
}

#ifdef NDZIP_STAGE_KERNELS
template<typename T, int Dims, bool Aligned>
hipError_t launch_debug_profile(int stage, const grid_geom &gg, uint32_t hc, const void *in, void *out, uint32_t *out_len,
        hipStream_t stream) {
    using W = typename word_of<T>::type;
    if constexpr (sizeof(W) == 8) {
        if (stage == debug_forward_transform || stage == debug_encode_residuals || stage == debug_decode_residuals_wide
                || stage == debug_inverse_transform_wide) {
            using L = wide::layout<W>;
            constexpr uint32_t enc = L::cube_bytes + L::zero_bytes + 64, dec = wide::decode_layout<Dims>::smem_bytes;
            const uint32_t smem = enc > dec ? enc : dec;
            hipLaunchKernelGGL((debug_stage_wide_kernel<Dims, Aligned>), dim3(1), dim3(wide::threads), smem, stream, stage, gg, hc,
                    static_cast<const W *>(in), static_cast<W *>(out), out_len);
            return hipGetLastError();
        }
    }
    using L = lds_layout<W>;
    const uint32_t smem = L::cube_bytes + L::zero_bytes + 64;
    hipLaunchKernelGGL((debug_stage_kernel<T, Dims, Aligned>), dim3(1), dim3(threads_per_hc), smem, stream, stage, gg, hc,
            static_cast<const W *>(in), static_cast<W *>(out), out_len);
    return hipGetLastError();
}

#endif  // NDZIP_STAGE_KERNELS

#define NDZIP_DISPATCH(FN, dims, aligned, ...)                                          \
    switch (dims) {                                                                     \
        case 1: return (aligned) ? FN<T_, 1, true>(__VA_ARGS__) : FN<T_, 1, false>(__VA_ARGS__); \
        case 2: return (aligned) ? FN<T_, 2, true>(__VA_ARGS__) : FN<T_, 2, false>(__VA_ARGS__); \
        case 3: return (aligned) ? FN<T_, 3, true>(__VA_ARGS__) : FN<T_, 3, false>(__VA_ARGS__); \
        default: return hipErrorInvalidValue;                                           \
    }

}  // namespace

#ifndef NDZIP_STAGE_KERNELS
template<>
int compress_hcs_per_group<T_>(int) {
    return tile_cfg<T_, 1>::K;
}

template<>
uint32_t compress_num_tiles<T_>(int, uint32_t nhc) {
    return (nhc + tile_cfg<T_, 1>::K - 1) / tile_cfg<T_, 1>::K;  // one descriptor per tile: K hypercubes (2 for 32-bit, 1 for 64-bit profiles)
}

template<>
hipError_t launch_compress<T_>(int dims, const compress_args &a) {
    NDZIP_DISPATCH(launch_compress_profile, dims, a.aligned, a)
}

template<>
hipError_t launch_decompress<T_>(int dims, const decompress_args &a) {
    NDZIP_DISPATCH(launch_decompress_profile, dims, a.aligned, a)
}

#else  // NDZIP_STAGE_KERNELS: the stage library (stages_f32.hip / stages_f64.hip) gets the stage entry point and nothing else
template<>
hipError_t launch_debug_stage<T_>(int stage, int dims, const grid_geom &gg, uint32_t hc, const void *in, void *out,
        uint32_t *out_len, uint32_t n, bool aligned, hipStream_t stream) {
    if (stage == debug_wave_scan) {
        if (n == 0 || n % 64 != 0) return hipErrorInvalidValue;
        hipLaunchKernelGGL(debug_wave_scan_kernel, dim3(n / 64), dim3(64), 0, stream, static_cast<const uint32_t *>(in),
                static_cast<uint32_t *>(out), n);
        return hipGetLastError();
    }
    if (stage == debug_lookback_scan) {
        // in: n lengths; out: n prefixes, the total, the error word.  Scratch of its own (zeroed once), two launches on it.
        if (n == 0) return hipErrorInvalidValue;
        int dev = 0, cus = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e == hipSuccess) e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        if (e != hipSuccess) return e;
        uint32_t grid = hc ? hc : static_cast<uint32_t>(cus) * 4u;  // (4 one-wavefront workgroups per CU are resident on any part)
        if (grid > n) grid = n;
        const uint32_t num_classes = grid < ticket_classes ? 1u : ticket_classes;
        tile_desc *scratch = nullptr;
        const size_t entries = static_cast<size_t>(n) + scratch_extra_descs;
        e = hipMalloc(reinterpret_cast<void **>(&scratch), entries * sizeof(tile_desc));
        if (e != hipSuccess) return e;
        e = hipMemsetAsync(scratch, 0, entries * sizeof(tile_desc), stream);
        uint32_t *o = static_cast<uint32_t *>(out);
        if (e == hipSuccess) e = hipMemsetAsync(o + n, 0, 2 * sizeof(uint32_t), stream);
        if (e == hipSuccess) e = init_scratch_epoch(scratch, n, stream);
        for (int launch = 0; launch < 2 && e == hipSuccess; ++launch) {  // (the second one runs under the epoch the first one left)
            hipLaunchKernelGGL(debug_lookback_kernel, dim3(grid), dim3(64), 0, stream, static_cast<const uint32_t *>(in), o, n,
                    scratch + scratch_extra_descs, reinterpret_cast<uint32_t *>(scratch + 16), num_classes, o + n, o + n + 1);
            e = hipGetLastError();
        }
        const hipError_t s = hipStreamSynchronize(stream);
        (void) hipFree(scratch);
        return e != hipSuccess ? e : s;
    }
    if (stage == debug_transpose32 || stage == debug_transpose32_generic) {
        if (n == 0) return hipSuccess;
        hipLaunchKernelGGL(debug_transpose_kernel, dim3((n + 63) / 64), dim3(64), 0, stream, static_cast<const uint32_t *>(in),
                static_cast<uint32_t *>(out), n, stage == debug_transpose32_generic ? 1 : 0);
        return hipGetLastError();
    }
    NDZIP_DISPATCH(launch_debug_profile, dims, aligned, stage, gg, hc, in, out, out_len, stream)
}

#endif  // NDZIP_STAGE_KERNELS

#undef NDZIP_DISPATCH

}  // namespace ndzip_hip
