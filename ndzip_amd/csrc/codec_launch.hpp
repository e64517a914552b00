// ndzip_amd/csrc/codec_launch.hpp -- host-visible launch interface between the C ABI (capi.hip) and the
// per-type kernel translation units (kernels_f32.hip / kernels_f64.hip).
#pragma once

#include <hip/hip_runtime.h>

#include "codec_common.hpp"

namespace ndzip_hip {

// Tile descriptor of the fused decoupled look-back scan: (status << 32) | value, written with ONE relaxed
// agent-scope 8-byte store, so the value is its own flag (no fences; MI355X guide section 6 G16 "R2").
using tile_desc = unsigned long long;

// Scratch in front of the descriptors: 16 experiment counters, then one ticket counter per class, each in its own 128-byte
// line, then the 'workgroups done' line (see release_tickets in codec_launch.inl), then the line of the launch epoch.  Sizes in
// tile_desc units.
constexpr unsigned ticket_classes = 16;       // ticket counters a launch uses (1 when the grid is smaller than that)
constexpr unsigned ticket_stride_words = 32;  // uint32 words between the counters of consecutive classes
constexpr unsigned scratch_extra_descs = 16 + (ticket_classes + 2) * ticket_stride_words / 2;
// The launch epoch lives in the scratch, not in a kernel argument: every workgroup reads it on its way in, the last one to leave
// writes the next one (release_tickets).  A launch therefore carries NO per-launch state from the host -- a compress call recorded
// into a hipGraph replays correctly (with the epoch as an argument every replay would run under the recorded epoch, find the previous
// replay's descriptors "published" and hand their stale lengths to the look-back: a silently wrong stream).
// uint32 index from the first ticket counter; [epoch_word + 1] = number of descriptors behind the scratch (what the last workgroup
// clears when the 30-bit epoch starts over).  The owner of the scratch sets both once (epoch 1).
constexpr unsigned epoch_word = (ticket_classes + 1) * ticket_stride_words;
constexpr unsigned epoch_limit = 1u << 30;

struct compress_args {
    const void *in;        // device, value_type[num_elements]
    grid_geom gg;
    uint32_t *header;      // device, NHC uint32 entries (+1 pad entry for 64-bit streams with odd NHC)
    void *body;            // device, first body word (= stream + header words for a contiguous stream)
    tile_desc *desc;       // device scratch, >= num_tiles + scratch_extra_descs entries, zeroed ONCE by its owner, who also sets the
                           // epoch line (init_scratch_epoch): the kernels advance the epoch themselves
    uint32_t *out_len;     // device scalar or nullptr
    uint32_t len_extra;    // header words + border words, added to the body length for *out_len
    uint32_t *err;         // device error word (sticky)
    hipStream_t stream;
    int num_cus;
    int device;            // HIP device ordinal the launch goes to (keys the per-device occupancy cache)
    int max_blocks_per_cu; // 0 = as many workgroups per CU as are resident; > 0 caps it (the relaunch after a look-back time-out runs
                           // one workgroup per CU: a grid a shared GPU is far more likely to hold in full)
    bool aligned;          // 16-byte aligned base and row strides
};

// Which kernel decodes a 64-bit hypercube when the caller has not chosen (ndzip_hip_decompressor_set_f64_work_items).  The
// 256-work-item decoder (decompress_kernel_wide: 72-82 VGPRs, 5-7 wavefronts per SIMD) is the one the register / occupancy numbers
// favour, but it has never executed on a GPU; until the -m gpu suite (test_f64_decoder_mappings_agree) and the A/B of
// tools/bench_configs.sh have run on an MI355X the default stays the 128-work-item mapping whose round-1 form was measured there.
// Flip this one constant when that evidence exists.
constexpr int default_f64_work_items = 128;

struct decompress_args {
    const uint32_t *header;   // NHC offset_after entries
    const uint32_t *header_base;  // device pointer to the value the entries are relative to (nullptr = 0; the global offset of
                              // `body`'s first word for a shard)
    const void *body;
    void *out;
    grid_geom gg;
    uint32_t *err;
    hipStream_t stream;
    bool aligned;
    uint32_t body_words;      // words of `body` the caller vouches for (hypercube runs + border); 0xffffffff = unknown
    int num_xcds;             // accelerator complexes (separate L2s) workgroups are dealt to round-robin: hipDeviceAttributeNumberOfXccs
    int f64_work_items;       // work-items per 64-bit hypercube: 256 = decompress_kernel_wide, 128 = decompress_kernel, 0 = default_f64_work_items
};

// hypercubes per compress / decompress workgroup for (T, dims)
template<typename T>
int compress_hcs_per_group(int dims);

template<typename T>
uint32_t compress_num_tiles(int dims, uint32_t nhc);

template<typename T>
hipError_t launch_compress(int dims, const compress_args &a);

// once, behind the zeroing of a scratch (stream-ordered): epoch 1, `desc_count` descriptors
inline hipError_t init_scratch_epoch(tile_desc *scratch, uint32_t desc_count, hipStream_t stream) {
    const uint32_t first_epoch = 1;
    uint32_t *line = reinterpret_cast<uint32_t *>(scratch + 16) + epoch_word;
    hipError_t e = hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(line), static_cast<int>(first_epoch), 1, stream);
    if (e == hipSuccess) e = hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(line + 1), static_cast<int>(desc_count), 1, stream);
    return e;
}

template<typename T>
hipError_t launch_decompress(int dims, const decompress_args &a);

// ---- stage entry points used by the parity tests (one hypercube, 128 work-items) -------------------------
enum debug_stage : int {
    debug_forward_transform = 0,  // in: array + geometry, hc index   -> out: 4096 residual words
    debug_encode_residuals = 1,   // in: 4096 residual words          -> out: encoded run, out_len = words
    debug_decode_residuals = 2,   // in: encoded run                  -> out: 4096 residual words
    debug_inverse_transform = 3,  // in: 4096 residual words          -> out: array (hypercube hc of geometry)
    debug_transpose32 = 4,        // in: n*32 uint32                  -> out: n*32 uint32 (v_perm network)
    debug_transpose32_generic = 5,
    debug_wave_scan = 6,          // in: n uint32 (n a multiple of 64)   -> out: per wavefront of 64, the inclusive prefix sums;
                                  //                                         out[n + w] = the wave sum of wavefront w
    debug_decode_residuals_wide = 8,  // stage 2 through the 256-work-item f64 decoder (wide::decode_residuals); f64 only
    debug_inverse_transform_wide = 9, // stage 3 through wide::inverse_transform; f64 only
    debug_lookback_scan = 7,      // in: n uint32 tile lengths           -> out: their n exclusive prefix sums, out[n] = the total,
                                  //     out[n + 1] = the error word; `hc` = workgroups of the persistent grid (0: as many as the
                                  //     production launch would use).  The production ticket / publish / look-back / release
                                  //     functions on their own, two launches on one scratch (the kernel's own epochs 1 and 2): the device-wide
                                  //     scan at tile counts no array that fits a test can reach
};

template<typename T>
hipError_t launch_debug_stage(int stage, int dims, const grid_geom &gg, uint32_t hc, const void *in, void *out,
        uint32_t *out_len, uint32_t n, bool aligned, hipStream_t stream);

// explicit specialisations live in kernels_f32.hip / kernels_f64.hip
#define NDZIP_DECLARE_LAUNCHERS(T)                                                                                     \
    template<> int compress_hcs_per_group<T>(int);                                                                    \
    template<> uint32_t compress_num_tiles<T>(int, uint32_t);                                                         \
    template<> hipError_t launch_compress<T>(int, const compress_args &);                                            \
    template<> hipError_t launch_decompress<T>(int, const decompress_args &);                                        \
    template<> hipError_t launch_debug_stage<T>(int, int, const grid_geom &, uint32_t, const void *, void *, uint32_t *, \
            uint32_t, bool, hipStream_t);
NDZIP_DECLARE_LAUNCHERS(float)
NDZIP_DECLARE_LAUNCHERS(double)
#undef NDZIP_DECLARE_LAUNCHERS

}  // namespace ndzip_hip
