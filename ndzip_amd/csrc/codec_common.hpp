// ndzip_amd/csrc/codec_common.hpp -- format constants and host/device geometry shared by every kernel.
//
// The stream format is the reference's (celerity/ndzip src/ndzip/common.hh:328-366);
// the decomposition below (128 work-items per 4096-element hypercube, 32 consecutive cube-local values per
// work-item, one 64-wide wavefront = 64 chunks of an f32 cube or 32 chunks of an f64 cube) is gfx950-specific
// and shares nothing with the reference's warp-32 layouts (src/ndzip/gpu_common.hh).
#pragma once

#include <cstdint>

namespace ndzip_hip {

constexpr int hc_size = 4096;          // elements per hypercube, all profiles (common.hh:368-381)
constexpr int threads_per_hc = 128;    // 2 wavefronts
constexpr int vals_per_thread = 32;    // cube-local values [32 t, 32 t + 32)
constexpr int max_dims = 3;

template<typename T>
struct word_of;
template<>
struct word_of<float> {
    using type = uint32_t;
};
template<>
struct word_of<double> {
    using type = uint64_t;
};

template<int Dims>
struct side_of;
template<>
struct side_of<1> {
    static constexpr uint32_t value = 4096;
};
template<>
struct side_of<2> {
    static constexpr uint32_t value = 64;
};
template<>
struct side_of<3> {
    static constexpr uint32_t value = 16;
};

template<typename T, int Dims>
struct profile {
    using value_type = T;
    using word = typename word_of<T>::type;               // bits_type of the reference (ndzip.hh:188-212)
    static constexpr int dims = Dims;
    static constexpr int B = sizeof(word) * 8;             // bits per word = values per chunk
    static constexpr uint32_t side = side_of<Dims>::value;
    static constexpr int head_words = hc_size / B;         // 128 (f32) / 64 (f64)
    static constexpr int max_hc_words = hc_size + head_words;  // compressed_block_length_bound (common.hh:391-392)
    static constexpr int words_per_header_word = B / 32;   // uint32 header entries per stream word
};

// Geometry of one array, passed by value to kernels.  Dimension d = 0 is slowest (ndzip.hh:172-180).
struct grid_geom {
    uint32_t n[max_dims];       // extent (unused leading entries = 1)
    uint32_t g[max_dims];       // hypercube grid floor(n / side)
    uint64_t stride[max_dims];  // element strides
    uint32_t g_magic[max_dims]; // floor(2^32 / g[d]) (0xffffffff for g = 1): division by g without a divider
    uint32_t nhc;               // num_hypercubes (common.hh:395-402)
    uint32_t dims;
};

inline uint32_t side_for_dims(int dims) { return dims == 1 ? 4096u : dims == 2 ? 64u : 16u; }

// Host-side: build geometry; dimension order preserved (n[0] slowest of the `dims` used entries).
inline grid_geom make_geom(int dims, const uint32_t *extent) {
    grid_geom gg{};
    gg.dims = static_cast<uint32_t>(dims);
    const uint32_t side = side_for_dims(dims);
    uint64_t nhc = 1;
    bool nhc_overflow = false;
    for (int d = 0; d < max_dims; ++d) {
        gg.n[d] = d < dims ? extent[d] : 1;
        gg.g[d] = d < dims ? extent[d] / side : 1;
        gg.g_magic[d] = gg.g[d] <= 1 ? 0xffffffffu : static_cast<uint32_t>((1ull << 32) / gg.g[d]);
        if (d < dims && __builtin_mul_overflow(nhc, static_cast<uint64_t>(gg.g[d]), &nhc)) nhc_overflow = true;
    }
    // saturates for extents beyond the format's limits (num_elements then exceeds 2^32 - 1 as well: every entry point
    // rejects such an extent before nhc is used)
    gg.nhc = nhc_overflow || nhc > 0xffffffffull ? 0xffffffffu : static_cast<uint32_t>(nhc);
    uint64_t s = 1;
    for (int d = dims - 1; d >= 0; --d) {
        gg.stride[d] = s;
        s *= gg.n[d];
    }
    for (int d = dims; d < max_dims; ++d) gg.stride[d] = 0;
    return gg;
}

// product of the extents, saturating at 2^64 - 1 (three 32-bit extents can exceed 64 bits)
inline uint64_t num_elements(const grid_geom &gg) {
    uint64_t n = 1;
    for (uint32_t d = 0; d < gg.dims; ++d) {
        if (__builtin_mul_overflow(n, static_cast<uint64_t>(gg.n[d]), &n)) return ~0ull;
    }
    return n;
}

// border_element_count (common.hh:308-317)
inline uint64_t border_count(const grid_geom &gg) {
    uint64_t covered = 1;
    const uint32_t side = side_for_dims(static_cast<int>(gg.dims));
    for (uint32_t d = 0; d < gg.dims; ++d) covered *= static_cast<uint64_t>(gg.g[d]) * side;
    return num_elements(gg) - covered;
}

// Border elements in increasing global linear index form three piecewise-linear
// classes once the array is viewed as (nz, ny, nx) with leading size-1 / fully covered dimensions:
//   tails   z < Cz, y < Cy, x in [Cx, nx)         -- (nx - Cx) per row
//   y-slabs z < Cz, y >= Cy, all x                -- (ny - Cy) * nx per z
//   z-slab  z >= Cz                               -- one contiguous block at the end
// so the i-th border element is found by division, without the reference's border_map recursion
// (src/ndzip/gpu_common.hh:277-344).
struct border_geom {
    uint64_t nx, ny, nz;   // extents mapped to 3D
    uint64_t cx, cy, cz;   // covered = g * side
    uint64_t tail;         // nx - cx
    uint64_t per_z;        // cy * tail + (ny - cy) * nx
    uint64_t count;        // total border elements
};

inline border_geom make_border_geom(const grid_geom &gg) {
    border_geom b{};
    const uint32_t side = side_for_dims(static_cast<int>(gg.dims));
    uint64_t n3[3] = {1, 1, 1}, c3[3] = {1, 1, 1};
    for (uint32_t d = 0; d < gg.dims; ++d) {
        n3[3 - gg.dims + d] = gg.n[d];
        c3[3 - gg.dims + d] = static_cast<uint64_t>(gg.g[d]) * side;
    }
    b.nz = n3[0];
    b.ny = n3[1];
    b.nx = n3[2];
    b.cz = c3[0];
    b.cy = c3[1];
    b.cx = c3[2];
    b.tail = b.nx - b.cx;
    b.per_z = b.cy * b.tail + (b.ny - b.cy) * b.nx;
    b.count = b.cz * b.per_z + (b.nz - b.cz) * b.ny * b.nx;
    return b;
}

}  // namespace ndzip_hip
