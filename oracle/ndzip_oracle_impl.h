/*
 * oracle/ndzip_oracle_impl.h -- width-generic body of the CPU oracle (TEST INFRASTRUCTURE ONLY).
 *
 * Included twice by ndzip_oracle.c, once with W = uint32_t / B = 32 (float streams) and once with
 * W = uint64_t / B = 64 (double streams).  FN(x) appends the width suffix.
 *
 * Every function restates -- it does not copy -- one function of the reference CPU path; the
 * reference location it follows is cited as file:line relative to /root/reference.
 */

#ifndef W
#error "include from ndzip_oracle.c only"
#endif

/* rotate_left_1 / rotate_right_1 / complement_negative: src/ndzip/common.hh:436-449 */
static inline W FN(rotl1)(W v) { return (W) ((v << 1) | (v >> (B - 1))); }
static inline W FN(rotr1)(W v) { return (W) ((v >> 1) | (v << (B - 1))); }
static inline W FN(compl_neg)(W v) { return (v >> (B - 1)) ? (W) (v ^ ((W) ~(W) 0 >> 1)) : v; }

/* load_hypercube: src/ndzip/cpu_codec.inl:74-85 with for_each_hypercube_slice common.hh:538-568.
 * Gathers the 4096 elements of hypercube `hc` (row-major over the hypercube grid, common.hh:414-433)
 * in cube-local row-major order. */
static void FN(load_cube)(const W *data, const geom *g, uint32_t hc, W *cube) {
    uint32_t c[3];
    geom_hc_origin(g, hc, c);
    const uint32_t s = g->side;
    if (g->dims == 1) {
        memcpy(cube, data + c[0], (size_t) s * sizeof(W));
    } else if (g->dims == 2) {
        for (uint32_t i = 0; i < s; ++i) {
            memcpy(cube + (size_t) i * s, data + (size_t) (c[0] + i) * g->n[1] + c[1], (size_t) s * sizeof(W));
        }
    } else {
        for (uint32_t i = 0; i < s; ++i) {
            for (uint32_t j = 0; j < s; ++j) {
                memcpy(cube + ((size_t) i * s + j) * s,
                        data + ((size_t) (c[0] + i) * g->n[1] + (c[1] + j)) * g->n[2] + c[2], (size_t) s * sizeof(W));
            }
        }
    }
}

/* store_hypercube: src/ndzip/cpu_codec.inl:87-98 */
static void FN(store_cube)(W *data, const geom *g, uint32_t hc, const W *cube) {
    uint32_t c[3];
    geom_hc_origin(g, hc, c);
    const uint32_t s = g->side;
    if (g->dims == 1) {
        memcpy(data + c[0], cube, (size_t) s * sizeof(W));
    } else if (g->dims == 2) {
        for (uint32_t i = 0; i < s; ++i) {
            memcpy(data + (size_t) (c[0] + i) * g->n[1] + c[1], cube + (size_t) i * s, (size_t) s * sizeof(W));
        }
    } else {
        for (uint32_t i = 0; i < s; ++i) {
            for (uint32_t j = 0; j < s; ++j) {
                memcpy(data + ((size_t) (c[0] + i) * g->n[1] + (c[1] + j)) * g->n[2] + c[2],
                        cube + ((size_t) i * s + j) * s, (size_t) s * sizeof(W));
            }
        }
    }
}

/* block_transform: generic version src/ndzip/common.hh:451-501.  rotl1 on every element, then a
 * first-order difference along every axis (first element of each line kept), then complement_negative.
 * The per-axis operators commute, so we run them in axis order fastest->slowest;
 * differences are taken back-to-front so that each step reads the not-yet-modified predecessor. */
EXPORT void FN(ndzip_oracle_forward_transform)(W *x, int dims) {
    const uint32_t s = side_of_dims(dims);
    for (uint32_t i = 0; i < HC_SIZE; ++i) x[i] = FN(rotl1)(x[i]);
    uint32_t stride = 1;
    for (int a = 0; a < dims; ++a, stride *= s) {
        /* lines along this axis: `outer` selects the slower coordinates, `inner` the faster ones */
        for (uint32_t outer = 0; outer < HC_SIZE; outer += stride * s) {
            for (uint32_t i = s - 1; i >= 1; --i) {
                W *cur = x + outer + i * stride;
                const W *prev = cur - stride;
                for (uint32_t inner = 0; inner < stride; ++inner) cur[inner] = (W) (cur[inner] - prev[inner]);
            }
        }
    }
    for (uint32_t i = 0; i < HC_SIZE; ++i) x[i] = FN(compl_neg)(x[i]);
}

/* inverse_block_transform: src/ndzip/common.hh:503-535.  complement_negative, inclusive prefix sum
 * along every axis, rotr1. */
EXPORT void FN(ndzip_oracle_inverse_transform)(W *x, int dims) {
    const uint32_t s = side_of_dims(dims);
    for (uint32_t i = 0; i < HC_SIZE; ++i) x[i] = FN(compl_neg)(x[i]);
    uint32_t stride = 1;
    for (int a = 0; a < dims; ++a, stride *= s) {
        for (uint32_t outer = 0; outer < HC_SIZE; outer += stride * s) {
            for (uint32_t i = 1; i < s; ++i) {
                W *cur = x + outer + i * stride;
                const W *prev = cur - stride;
                for (uint32_t inner = 0; inner < stride; ++inner) cur[inner] = (W) (cur[inner] + prev[inner]);
            }
        }
    }
    for (uint32_t i = 0; i < HC_SIZE; ++i) x[i] = FN(rotr1)(x[i]);
}

/* transpose_bits_trivial: src/ndzip/cpu_codec.inl:355-363.  out[i] bit (B-1-j) = in[j] bit (B-1-i).
 * Kept as the definition; the fast version below is checked against it in tests/test_oracle.py. */
EXPORT void FN(ndzip_oracle_transpose_bits_trivial)(const W *in, W *out) {
    for (unsigned i = 0; i < B; ++i) {
        W o = 0;
        for (unsigned j = 0; j < B; ++j) { o |= (W) (((in[j] >> (B - 1 - i)) & 1u) << (B - 1 - j)); }
        out[i] = o;
    }
}

/* Same function as a log2(B)-stage block-swap network (the role transpose_bits_avx2 plays in the
 * reference, cpu_codec.inl:367-501, though not its algorithm): viewing word r as matrix row r with the MSB
 * as column 0, stage s swaps the off-diagonal s x s blocks of every 2s x 2s diagonal block. */
#define SWAP_STAGE(S, M)                                           \
    for (unsigned r = 0; r < B; ++r) {                             \
        if (r & (S)) continue;                                     \
        const W a = x[r], b = x[r + (S)];                          \
        x[r] = (W) ((a & ~(W) (M)) | ((b >> (S)) & (W) (M)));      \
        x[r + (S)] = (W) (((a << (S)) & ~(W) (M)) | (b & (W) (M))); \
    }
EXPORT void FN(ndzip_oracle_transpose_bits)(const W *in, W *out) {
    W x[B];
    memcpy(x, in, sizeof x);
    /* mask M of stage S: bit positions p with (p & S) == 0 */
    if (B == 64) { SWAP_STAGE(32 % B, 0x00000000FFFFFFFFull) }
    SWAP_STAGE(16, 0x0000FFFF0000FFFFull)
    SWAP_STAGE(8, 0x00FF00FF00FF00FFull)
    SWAP_STAGE(4, 0x0F0F0F0F0F0F0F0Full)
    SWAP_STAGE(2, 0x3333333333333333ull)
    SWAP_STAGE(1, 0x5555555555555555ull)
    memcpy(out, x, sizeof x);
}
#undef SWAP_STAGE

/* zero_bit_encode: src/ndzip/cpu_codec.inl:541-559 (generate_zero_map :344-352, compact_zero_words
 * :514-524).  Heads of all chunks first, then the non-zero bit planes of every chunk in chunk order.
 * Returns the number of W words written. */
EXPORT uint32_t FN(ndzip_oracle_encode_cube)(const W *cube, W *out) {
    uint32_t head_pos = 0;
    uint32_t body_pos = HC_SIZE / B;
    for (uint32_t off = 0; off < HC_SIZE; off += B) {
        W head = 0;
        for (unsigned j = 0; j < B; ++j) head |= cube[off + j];
        out[head_pos++] = head;
        if (head != 0) {
            W planes[B];
            FN(ndzip_oracle_transpose_bits)(cube + off, planes);
            for (unsigned i = 0; i < B; ++i) {
                if (planes[i] != 0) out[body_pos++] = planes[i];
            }
        }
    }
    return body_pos;
}

/* zero_bit_decode: src/ndzip/cpu_codec.inl:561-578 (expand_zero_words :526-538).
 * Returns the number of W words consumed. */
EXPORT uint32_t FN(ndzip_oracle_decode_cube)(const W *in, W *cube) {
    uint32_t head_pos = 0;
    uint32_t body_pos = HC_SIZE / B;
    for (uint32_t off = 0; off < HC_SIZE; off += B) {
        const W head = in[head_pos++];
        if (head == 0) {
            memset(cube + off, 0, B * sizeof(W));
        } else {
            W planes[B];
            for (unsigned i = 0; i < B; ++i) {
                planes[i] = ((head >> (B - 1 - i)) & 1u) ? in[body_pos++] : (W) 0;
            }
            FN(ndzip_oracle_transpose_bits)(planes, cube + off);
        }
    }
    return body_pos;
}

/* compressed_length_bound: src/ndzip/common.cc:31-55 */
static uint64_t FN(length_bound)(const geom *g) {
    const uint64_t header = ((uint64_t) g->nhc + (B / 32) - 1) / (B / 32);
    return header + (uint64_t) g->nhc * (HC_SIZE / B * (B + 1)) + geom_border_count(g);
}

/* stream<Profile>::hypercube(0) - buffer: src/ndzip/common.hh:350-358 */
static inline uint32_t FN(header_words)(uint32_t nhc) { return (nhc + (B / 32) - 1) / (B / 32); }

/* pack_border / unpack_border: src/ndzip/common.hh:284-306 over for_each_border_slice :245-282.
 * Border elements are appended verbatim in increasing global linear index. */
static uint32_t FN(pack_border)(W *dst, const W *src, const geom *g) {
    uint32_t n = 0;
    slice_iter it;
    slice_iter_init(&it, g);
    uint64_t off;
    uint32_t cnt;
    while (slice_iter_next(&it, &off, &cnt)) {
        memcpy(dst + n, src + off, (size_t) cnt * sizeof(W));
        n += cnt;
    }
    return n;
}

static uint32_t FN(unpack_border)(W *dst, const W *src, const geom *g) {
    uint32_t n = 0;
    slice_iter it;
    slice_iter_init(&it, g);
    uint64_t off;
    uint32_t cnt;
    while (slice_iter_next(&it, &off, &cnt)) {
        memcpy(dst + off, src + n, (size_t) cnt * sizeof(W));
        n += cnt;
    }
    return n;
}

/* serial_compressor::compress: src/ndzip/cpu_codec.inl:597-619.
 * With num_threads > 1 this is a two-phase OpenMP restatement of openmp_compressor::compress
 * (:780-887): the reference resolves the in-order stream assembly with a priority queue of write
 * buffers; here every hypercube is encoded into a bound-sized scratch slot, lengths are prefix-summed
 * serially and bodies are copied in parallel.  Same per-hypercube arithmetic, same stream. */
EXPORT uint64_t FN(ndzip_oracle_compress)(int dims, const uint32_t *extent, const W *data, W *stream, int num_threads) {
    geom g;
    if (geom_init(&g, dims, extent) != 0) return 0;
    const uint32_t hw = FN(header_words)(g.nhc);
    uint32_t *header = (uint32_t *) stream;
    W *body = stream + hw;
    uint64_t offset = 0;

    if (num_threads <= 1 || g.nhc < 2) {
        W cube[HC_SIZE];
        for (uint32_t hc = 0; hc < g.nhc; ++hc) {
            FN(load_cube)(data, &g, hc, cube);
            FN(ndzip_oracle_forward_transform)(cube, dims);
            offset += FN(ndzip_oracle_encode_cube)(cube, body + offset);
            header[hc] = (uint32_t) offset;
        }
    } else {
#ifdef _OPENMP
        const uint32_t slot = HC_SIZE / B * (B + 1);
        /* process in batches so scratch stays bounded (reference: 30 write buffers, cpu_codec.inl:712) */
        const uint32_t batch = 4096;
        /* scratch is kept across calls (first-touch page faults of 35-70 MB per call would dominate the timing) */
        static W *scratch = NULL;
        static uint32_t *len = NULL;
        static uint64_t *start = NULL;
        if (!scratch) {
            scratch = (W *) malloc((size_t) batch * slot * sizeof(W));
            len = (uint32_t *) malloc((size_t) batch * sizeof(uint32_t));
            start = (uint64_t *) malloc((size_t) batch * sizeof(uint64_t));
        }
        /* ONE parallel region for the whole call (a region per batch means waking the team dozens of times) */
        uint64_t running = 0;
#pragma omp parallel num_threads(num_threads)
        {
            W cube[HC_SIZE];
            for (uint32_t first = 0; first < g.nhc; first += batch) {
                const uint32_t count = g.nhc - first < batch ? g.nhc - first : batch;
#pragma omp for schedule(dynamic, 4)
                for (uint32_t i = 0; i < count; ++i) {
                    FN(load_cube)(data, &g, first + i, cube);
                    FN(ndzip_oracle_forward_transform)(cube, dims);
                    len[i] = FN(ndzip_oracle_encode_cube)(cube, scratch + (size_t) i * slot);
                }
                /* in-order stream assembly: the serial dependency the reference resolves with its write queue */
#pragma omp single
                for (uint32_t i = 0; i < count; ++i) {
                    start[i] = running;
                    running += len[i];
                    header[first + i] = (uint32_t) running;
                }
#pragma omp for schedule(static)
                for (uint32_t i = 0; i < count; ++i) {
                    memcpy(body + start[i], scratch + (size_t) i * slot, (size_t) len[i] * sizeof(W));
                }
            }
        }
        offset = running;
#else
        return 0;
#endif
    }
    /* 64-bit streams with an odd hypercube count leave one uint32 of header padding; the reference
     * GPU paths zero it (cuda_codec.inl:446-452), the CPU path leaves the caller's buffer content. We
     * write 0 so the stream is deterministic. */
    if ((B == 64) && (g.nhc & 1u)) header[g.nhc] = 0;
    const uint32_t nb = FN(pack_border)(body + offset, data, &g);
    return (uint64_t) hw + offset + nb;
}

/* serial_decompressor::decompress: src/ndzip/cpu_codec.inl:640-659; OpenMP variant :890-923 is a
 * plain parallel-for over hypercubes since offsets come from the header. */
EXPORT uint64_t FN(ndzip_oracle_decompress)(int dims, const uint32_t *extent, const W *stream, W *data, int num_threads) {
    geom g;
    if (geom_init(&g, dims, extent) != 0) return 0;
    const uint32_t hw = FN(header_words)(g.nhc);
    const uint32_t *header = (const uint32_t *) stream;
    const W *body = stream + hw;
    (void) num_threads;
#ifdef _OPENMP
#pragma omp parallel num_threads(num_threads > 1 ? num_threads : 1)
#endif
    {
        W cube[HC_SIZE];
#ifdef _OPENMP
#pragma omp for schedule(static)
#endif
        for (uint32_t hc = 0; hc < g.nhc; ++hc) {
            const uint32_t begin = hc ? header[hc - 1] : 0;
            FN(ndzip_oracle_decode_cube)(body + begin, cube);
            FN(ndzip_oracle_inverse_transform)(cube, dims);
            FN(store_cube)(data, &g, hc, cube);
        }
    }
    const uint32_t end = g.nhc ? header[g.nhc - 1] : 0;
    const uint32_t nb = FN(unpack_border)(data, body + end, &g);
    return (uint64_t) hw + end + nb;
}

/* transform + encode one gathered cube: the per-hypercube unit the GPU stage tests compare against
 * (reference tests src/test/codec_profile_test.inl:552-729, :889-947). */
EXPORT void FN(ndzip_oracle_load_cube)(int dims, const uint32_t *extent, const W *data, uint32_t hc, W *cube) {
    geom g;
    if (geom_init(&g, dims, extent) != 0) return;
    FN(load_cube)(data, &g, hc, cube);
}
