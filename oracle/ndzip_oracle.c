/*
 * oracle/ndzip_oracle.c -- CPU oracle for the ndzip block encode/decode path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is a plain-C restatement of the reference's serial CPU algorithm
 * (celerity/ndzip, src/ndzip/common.hh + cpu_codec.inl).  It exists to CHECK the HIP product path and to
 * provide the `cpu_baseline` leg of bench.py.  Nothing under ndzip_amd/ may import, link or call it;
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do.
 *
 * Pinning: tests/test_oracle.py checks this oracle bit-for-bit against
 *   (a) the known-answer vectors the reference's own tests hold (for_each_border_slice lists,
 *       src/test/codec_generic_test.cc:102-111) and
 *   (b) streams produced by the reference itself, compiled from /root/reference by oracle/Makefile into
 *       oracle/_ref/libndzip_ref.so (fixtures under tests/golden/, generator tests/golden/make_golden.py),
 * and, when oracle/_ref is present, directly against the reference library on random inputs.
 *
 * All arithmetic is unsigned integer / bitwise on IEEE bit patterns; there is no floating point here.
 */

#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define EXPORT __attribute__((visibility("default")))
#define HC_SIZE 4096u

/* hypercube_side_length: src/ndzip/common.hh:368-381 */
static inline uint32_t side_of_dims(int dims) { return dims == 1 ? 4096u : dims == 2 ? 64u : 16u; }

typedef struct {
    int dims;
    uint32_t n[3];   /* extent, n[dims-1] fastest */
    uint32_t g[3];   /* hypercube grid = floor(n / side) */
    uint32_t side;
    uint32_t nhc;    /* num_hypercubes: src/ndzip/common.hh:395-402 */
    uint64_t nelem;
} geom;

static int geom_init(geom *g, int dims, const uint32_t *extent) {
    if (dims < 1 || dims > 3) return -1;
    g->dims = dims;
    g->side = side_of_dims(dims);
    g->nhc = 1;
    g->nelem = 1;
    for (int d = 0; d < 3; ++d) {
        g->n[d] = d < dims ? extent[d] : 1;
        g->g[d] = d < dims ? extent[d] / g->side : 1;
        if (d < dims) {
            g->nhc *= g->g[d];
            g->nelem *= g->n[d];
        }
    }
    return 0;
}

/* extent_from_linear_id(hc, size / side) * side: src/ndzip/common.hh:570-579, used at cpu_codec.inl:858-859;
 * identical to the nested-loop order of for_each_hypercube (common.hh:414-433). */
static void geom_hc_origin(const geom *g, uint32_t hc, uint32_t *c) {
    for (int nd = 0; nd < g->dims; ++nd) {
        const int d = g->dims - 1 - nd;
        c[d] = (hc % g->g[d]) * g->side;
        hc /= g->g[d];
    }
}

/* border_element_count: src/ndzip/common.hh:308-317 */
static uint64_t geom_border_count(const geom *g) {
    uint64_t cube_elems = 1;
    for (int d = 0; d < g->dims; ++d) cube_elems *= (uint64_t) g->g[d] * g->side;
    return g->nelem - cube_elems;
}

/* for_each_border_slice: src/ndzip/common.hh:245-282, restated as an iterator that walks "rows" (all
 * coordinates but the fastest) in increasing linear order: a row whose leading coordinates leave the
 * hypercube-covered region is border as a whole, otherwise only its tail past the last full hypercube
 * is.  Adjacent slices are merged so the emitted list equals the reference's slice list. */
typedef struct {
    const geom *g;
    uint64_t row, nrows;
    int whole_array;
    int done;
    int have_pending;
    int pend_kind;
    uint64_t pend_off;
    uint64_t pend_cnt;
} slice_iter;

static void slice_iter_init(slice_iter *it, const geom *g) {
    it->g = g;
    it->row = 0;
    it->nrows = 1;
    for (int d = 0; d + 1 < g->dims; ++d) it->nrows *= g->n[d];
    it->whole_array = 0;
    for (int d = 0; d < g->dims; ++d) {
        if (g->g[d] == 0) it->whole_array = 1;
    }
    it->done = 0;
    it->have_pending = 0;
    it->pend_kind = 0;
    it->pend_off = it->pend_cnt = 0;
}

/* `kind` = the slowest dimension whose coordinate leaves the covered region (dims-1 for a row tail);
 * the reference emits one slice per recursion level, so only same-kind neighbours are merged. */
static int slice_iter_row(const slice_iter *it, uint64_t row, uint64_t *off, uint64_t *cnt, int *kind) {
    const geom *g = it->g;
    const uint32_t last = g->n[g->dims - 1];
    const uint32_t covered_last = g->g[g->dims - 1] * g->side;
    int lead_border = 0;
    uint64_t r = row;
    for (int d = g->dims - 2; d >= 0; --d) {
        const uint32_t coord = (uint32_t) (r % g->n[d]);
        r /= g->n[d];
        if (coord >= g->g[d] * g->side) {
            lead_border = 1;
            *kind = d;
        }
    }
    if (lead_border) {
        *off = row * last;
        *cnt = last;
        return 1;
    }
    if (covered_last < last) {
        *off = row * last + covered_last;
        *cnt = last - covered_last;
        *kind = g->dims - 1;
        return 1;
    }
    return 0;
}

static int slice_iter_next(slice_iter *it, uint64_t *off, uint32_t *cnt) {
    if (it->done) return 0;
    if (it->whole_array) {
        it->done = 1;
        if (it->g->nelem == 0) return 0;
        *off = 0;
        *cnt = (uint32_t) it->g->nelem;
        return 1;
    }
    for (;;) {
        if (it->row >= it->nrows) {
            it->done = 1;
            if (it->have_pending) {
                *off = it->pend_off;
                *cnt = (uint32_t) it->pend_cnt;
                it->have_pending = 0;
                return 1;
            }
            return 0;
        }
        uint64_t o, c;
        int kind = 0;
        const int has = slice_iter_row(it, it->row, &o, &c, &kind);
        it->row++;
        if (!has) continue;
        if (it->have_pending && it->pend_kind == kind && kind != it->g->dims - 1
                && it->pend_off + it->pend_cnt == o) {
            it->pend_cnt += c;
            continue;
        }
        if (it->have_pending) {
            *off = it->pend_off;
            *cnt = (uint32_t) it->pend_cnt;
            it->pend_off = o;
            it->pend_cnt = c;
            it->pend_kind = kind;
            return 1;
        }
        it->have_pending = 1;
        it->pend_off = o;
        it->pend_cnt = c;
        it->pend_kind = kind;
    }
}

#define CAT_(a, b) a##b
#define CAT(a, b) CAT_(a, b)

#define W uint32_t
#define B 32u
#define FN(name) CAT(name, _u32)
#include "ndzip_oracle_impl.h"
#undef W
#undef B
#undef FN

#define W uint64_t
#define B 64u
#define FN(name) CAT(name, _u64)
#include "ndzip_oracle_impl.h"
#undef W
#undef B
#undef FN

EXPORT uint32_t ndzip_oracle_num_hypercubes(int dims, const uint32_t *extent) {
    geom g;
    if (geom_init(&g, dims, extent) != 0) return 0;
    return g.nhc;
}

EXPORT uint64_t ndzip_oracle_border_count(int dims, const uint32_t *extent) {
    geom g;
    if (geom_init(&g, dims, extent) != 0) return 0;
    return geom_border_count(&g);
}

/* compressed_length_bound<T>: src/ndzip/common.cc:44-55; bits = 32 | 64 */
EXPORT uint64_t ndzip_oracle_compressed_length_bound(int bits, int dims, const uint32_t *extent) {
    geom g;
    if (geom_init(&g, dims, extent) != 0) return 0;
    return bits == 32 ? length_bound_u32(&g) : length_bound_u64(&g);
}

/* Dump the border slice list (offset,count pairs) -- checked against the reference's known answers
 * src/test/codec_generic_test.cc:102-111.  `side` overrides the profile side length like the test does.
 * Returns the number of slices; writes at most `cap` pairs. */
EXPORT uint32_t ndzip_oracle_border_slices(int dims, const uint32_t *extent, uint32_t side, uint64_t *pairs, uint32_t cap) {
    geom g;
    if (geom_init(&g, dims, extent) != 0) return 0;
    if (side) {
        g.side = side;
        g.nhc = 1;
        for (int d = 0; d < dims; ++d) {
            g.g[d] = extent[d] / side;
            g.nhc *= g.g[d];
        }
    }
    slice_iter it;
    slice_iter_init(&it, &g);
    uint32_t n = 0;
    uint64_t off;
    uint32_t cnt;
    while (slice_iter_next(&it, &off, &cnt)) {
        if (n < cap) {
            pairs[2 * n] = off;
            pairs[2 * n + 1] = cnt;
        }
        ++n;
    }
    return n;
}

/* First-touch helper for timing runs on multi-socket hosts: copies (src != NULL) or zero-fills `bytes` at `dst` in
 * 2 MiB blocks spread over the OpenMP team, so the pages end up distributed over the NUMA nodes instead of all on the
 * node of the Python main thread. */
EXPORT void ndzip_oracle_parallel_copy(void *dst, const void *src, uint64_t bytes, int num_threads) {
    const uint64_t block = 2u << 20;
    const uint64_t nblocks = (bytes + block - 1) / block;
    (void) num_threads;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(num_threads > 1 ? num_threads : 1)
#endif
    for (uint64_t b = 0; b < nblocks; ++b) {
        const uint64_t off = b * block;
        const uint64_t n = bytes - off < block ? bytes - off : block;
        if (src) {
            memcpy((char *) dst + off, (const char *) src + off, n);
        } else {
            memset((char *) dst + off, 0, n);
        }
    }
}

EXPORT int ndzip_oracle_max_threads(void) {
#ifdef _OPENMP
    extern int omp_get_max_threads(void);
    return omp_get_max_threads();
#else
    return 1;
#endif
}
