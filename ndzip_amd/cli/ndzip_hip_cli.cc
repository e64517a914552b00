// ndzip-hip -- compress or decompress binary float dumps on an MI355X through the C ABI of libndzip_hip.so.
//
// File-level drop-in for the reference's `compress` tool (src/compress/compress.cc): same options, same file format --
// the input is cut into arrays of `-n` elements, each becomes one ndzip stream, the output is their plain concatenation
// with no container header (compress.cc:17-58); decompression needs -n and -t again and walks the concatenation
// (compress.cc:61-86).  Files written by either tool are read by the other.
//
// Differences, on purpose:
//   * `-e` accepts `hip` only.  This back-end has no CPU path (use the reference tool with `-e cpu`); `-T` is accepted for
//     command-line compatibility and ignored.
//   * the chunk loop is pipelined: `--slots` arrays are in flight (H2D of chunk j+1, kernels of chunk j and D2H of chunk
//     j-1 overlap) through ndzip_hip_offloader_*; the reference's loop is one blocking offloader call per chunk.
//   * the summary line reports the true raw size (the reference multiplies by the chunk count twice, compress.cc:50-51).
//   * a trailing partial array is an error for compression, as in the reference (io.cc read_exact).
#include <cerrno>
#include <chrono>
#include <cinttypes>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/ndzip_hip.h"

namespace {

struct options {
    bool decompress = false;
    std::vector<uint32_t> size;
    int dtype = NDZIP_HIP_F32;
    std::string input = "-";
    std::string output = "-";
    int slots = 3;
    bool quiet = false;
};

[[noreturn]] void usage_error(const std::string &msg, const char *argv0) {
    fprintf(stderr, "%s\n\nUsage: %s [options]\n\n", msg.c_str(), argv0);
    fprintf(stderr,
            "Options:\n"
            "  --help                    show this help\n"
            "  -d [ --decompress ]       decompress (default compress)\n"
            "  -n [ --array-size ] arg   array size (one value per dimension, first-major)\n"
            "  -t [ --data-type ] arg    float|double (default float)\n"
            "  -e [ --target_str ] arg   hip (default hip)\n"
            "  -T [ --threads ] arg      ignored (no CPU path in this back-end)\n"
            "  -i [ --input ] arg        input file (default '-' is stdin)\n"
            "  -o [ --output ] arg       output file (default '-' is stdout)\n"
            "  --slots arg               arrays in flight on the device (default 3)\n"
            "  --no-mmap                 accepted, ignored (I/O is buffered through pinned memory)\n"
            "  -q [ --quiet ]            no summary line\n");
    exit(EXIT_FAILURE);
}

void check(int status, const char *what) {
    if (status != NDZIP_HIP_OK) throw std::runtime_error(std::string(what) + ": " + ndzip_hip_last_error());
}

uint32_t parse_u32(const std::string &s, const char *argv0) {
    char *end = nullptr;
    errno = 0;
    const unsigned long long v = strtoull(s.c_str(), &end, 10);
    if (s.empty() || *end || errno || v > 0xffffffffull) usage_error("the argument ('" + s + "') for option '--array-size' is invalid", argv0);
    return static_cast<uint32_t>(v);
}

options parse(int argc, char **argv) {
    options o;
    auto is_number = [](const char *s) { return *s && strspn(s, "0123456789") == strlen(s); };
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        auto value = [&](const char *name) -> std::string {
            if (i + 1 >= argc) usage_error(std::string("the required argument for option '") + name + "' is missing", argv[0]);
            return argv[++i];
        };
        if (a == "--help") {
            printf("Compress or decompress binary float dump\n\n");
            fflush(stdout);
            usage_error("", argv[0]);
        } else if (a == "-d" || a == "--decompress") {
            o.decompress = true;
        } else if (a == "-n" || a == "--array-size") {
            o.size.push_back(parse_u32(value("--array-size"), argv[0]));
            while (i + 1 < argc && is_number(argv[i + 1])) o.size.push_back(parse_u32(argv[++i], argv[0]));  // multitoken
        } else if (a == "-t" || a == "--data-type") {
            const std::string t = value("--data-type");
            if (t == "float") o.dtype = NDZIP_HIP_F32;
            else if (t == "double") o.dtype = NDZIP_HIP_F64;
            else usage_error("Invalid data type " + t, argv[0]);
        } else if (a == "-e" || a == "--target_str") {
            const std::string t = value("--target_str");
            if (t != "hip") usage_error("Unimplemented target " + t, argv[0]);
        } else if (a == "-T" || a == "--threads") {
            (void) value("--threads");
        } else if (a == "-i" || a == "--input") {
            o.input = value("--input");
        } else if (a == "-o" || a == "--output") {
            o.output = value("--output");
        } else if (a == "--slots") {
            o.slots = static_cast<int>(parse_u32(value("--slots"), argv[0]));
        } else if (a == "--no-mmap") {
        } else if (a == "-q" || a == "--quiet") {
            o.quiet = true;
        } else {
            usage_error("unrecognised option '" + a + "'", argv[0]);
        }
    }
    if (o.size.empty()) usage_error("the option '--array-size' is required but missing", argv[0]);
    if (o.size.size() > 3) usage_error("Expected between 1 and 3 dimensions, got " + std::to_string(o.size.size()), argv[0]);
    if (o.slots < 1 || o.slots > 16) usage_error("--slots must be between 1 and 16", argv[0]);
    return o;
}

struct file {
    FILE *f = nullptr;
    bool owned = false;
    file(const std::string &name, bool write) {
        if (!name.empty() && name != "-") {
            f = fopen(name.c_str(), write ? "wb" : "rb");
            if (!f) throw std::runtime_error("fopen: " + name + ": " + strerror(errno));
            owned = true;
        } else {
            f = write ? stdout : stdin;
        }
    }
    ~file() {
        if (owned && f) fclose(f);
    }
    size_t read(void *dst, size_t bytes) {
        const size_t n = fread(dst, 1, bytes, f);
        if (n < bytes && ferror(f)) throw std::runtime_error(std::string("fread: ") + strerror(errno));
        return n;
    }
    void write(const void *src, size_t bytes) {
        if (bytes && fwrite(src, bytes, 1, f) < 1) throw std::runtime_error(std::string("fwrite: ") + strerror(errno));
    }
    void flush() {
        if (fflush(f) != 0) throw std::runtime_error(std::string("fflush: ") + strerror(errno));
    }
};

struct pinned {
    void *p = nullptr;
    explicit pinned(size_t bytes) { check(ndzip_hip_host_alloc(bytes, &p), "pinned allocation"); }
    ~pinned() { ndzip_hip_host_free(p); }
    pinned(const pinned &) = delete;
    pinned &operator=(const pinned &) = delete;
};

struct offloader {
    ndzip_hip_offloader *h = nullptr;
    offloader(int dtype, int dims, const uint32_t *extent, int slots) { check(ndzip_hip_offloader_create(dtype, dims, extent, slots, &h), "creating the offloader"); }
    ~offloader() { ndzip_hip_offloader_destroy(h); }
};

// `slots` jobs in flight, retired in submission order
struct pipeline {
    int slots;
    size_t submitted = 0, retired = 0;
    explicit pipeline(int n) : slots(n) {}
    bool full() const { return submitted - retired == static_cast<size_t>(slots); }
    bool empty() const { return submitted == retired; }
    int next_slot() const { return static_cast<int>(submitted % slots); }
    int oldest_slot() const { return static_cast<int>(retired % slots); }
};

struct job_buffers {
    std::vector<std::unique_ptr<pinned>> in, out;
    job_buffers(int slots, size_t in_bytes, size_t out_bytes) {
        for (int s = 0; s < slots; ++s) {
            in.emplace_back(new pinned(in_bytes));
            out.emplace_back(new pinned(out_bytes));
        }
    }
};

struct sizes {
    int dims;
    size_t wb, chunk_bytes, bound_bytes, header_bytes;
    uint64_t bound_words;
};

sizes sizes_of(const options &o) {
    sizes z{};
    z.dims = static_cast<int>(o.size.size());
    z.wb = o.dtype == NDZIP_HIP_F32 ? 4 : 8;
    uint64_t n = 1;
    for (uint32_t e : o.size) n *= e;
    if (n == 0) throw std::runtime_error("array size has zero elements");
    check(ndzip_hip_compressed_length_bound(o.dtype, z.dims, o.size.data(), &z.bound_words), "compressed_length_bound");
    uint32_t nhc = 0, header_words = 0;
    check(ndzip_hip_num_hypercubes(z.dims, o.size.data(), &nhc), "num_hypercubes");
    check(ndzip_hip_header_words(o.dtype, nhc, &header_words), "header_words");
    z.chunk_bytes = static_cast<size_t>(n) * z.wb;
    z.bound_bytes = static_cast<size_t>(z.bound_words) * z.wb;
    z.header_bytes = static_cast<size_t>(header_words) * z.wb;
    return z;
}

// compress.cc:17-58
void compress_file(const options &o) {
    const sizes z = sizes_of(o);
    file in(o.input, false), out(o.output, true);
    offloader off(o.dtype, z.dims, o.size.data(), o.slots);
    job_buffers buf(o.slots, z.chunk_bytes, z.bound_bytes);
    pipeline pipe(o.slots);
    size_t compressed_words = 0;
    uint64_t kernel_ns = 0;
    const auto t0 = std::chrono::steady_clock::now();
    auto retire = [&] {
        const int slot = pipe.oldest_slot();
        uint32_t words = 0;
        uint64_t ns = 0;
        check(ndzip_hip_offloader_wait(off.h, slot, &words, &ns), "compress");
        out.write(buf.out[slot]->p, static_cast<size_t>(words) * z.wb);
        compressed_words += words;
        kernel_ns += ns;
        ++pipe.retired;
    };
    for (;;) {
        if (pipe.full()) retire();
        const int slot = pipe.next_slot();
        const size_t got = in.read(buf.in[slot]->p, z.chunk_bytes);
        if (got == 0) break;
        if (got != z.chunk_bytes) throw std::runtime_error("Input file size is not a multiple of the chunk size");  // io.cc read_exact
        check(ndzip_hip_offloader_submit_compress(off.h, slot, o.size.data(), buf.in[slot]->p, buf.out[slot]->p), "compress");
        ++pipe.submitted;
    }
    while (!pipe.empty()) retire();
    out.flush();
    if (!o.quiet) {
        const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        const size_t n_chunks = pipe.submitted, raw = n_chunks * z.chunk_bytes, comp = compressed_words * z.wb;
        fprintf(stderr, "raw = %zu bytes", raw);
        if (n_chunks > 1) fprintf(stderr, " (%zu chunks à %zu bytes)", n_chunks, z.chunk_bytes);
        fprintf(stderr, ", compressed = %zu bytes, ratio = %.4f, time = %.3fs (device), %.3fs (wall)\n", comp,
                raw ? static_cast<double>(comp) / static_cast<double>(raw) : 0.0, static_cast<double>(kernel_ns) * 1e-9, wall);
    }
}

// compress.cc:61-86
void decompress_file(const options &o) {
    const sizes z = sizes_of(o);
    file in(o.input, false), out(o.output, true);
    offloader off(o.dtype, z.dims, o.size.data(), o.slots);
    job_buffers buf(o.slots, z.bound_bytes, z.chunk_bytes);
    pipeline pipe(o.slots);
    auto retire = [&] {
        const int slot = pipe.oldest_slot();
        check(ndzip_hip_offloader_wait(off.h, slot, nullptr, nullptr), "decompress");
        out.write(buf.out[slot]->p, z.chunk_bytes);
        ++pipe.retired;
    };
    for (;;) {
        if (pipe.full()) retire();
        const int slot = pipe.next_slot();
        // one stream = header (whose last entry gives the body length) + bodies + border: read the header, then the rest
        char *stream = static_cast<char *>(buf.in[slot]->p);
        const size_t got = in.read(stream, z.header_bytes);
        if (got == 0 && z.header_bytes > 0) break;
        if (got != z.header_bytes) throw std::runtime_error("truncated stream header in input");
        uint32_t words = 0;
        check(ndzip_hip_stream_words(o.dtype, z.dims, o.size.data(), stream, z.bound_words, &words), "stream header");
        const size_t rest = static_cast<size_t>(words) * z.wb - z.header_bytes;
        const size_t got_rest = in.read(stream + z.header_bytes, rest);
        if (z.header_bytes == 0 && got_rest == 0) break;  // array without hypercubes: the stream is the border alone
        if (got_rest != rest) throw std::runtime_error("truncated stream in input");
        check(ndzip_hip_offloader_submit_decompress(off.h, slot, o.size.data(), stream, words, buf.out[slot]->p), "decompress");
        ++pipe.submitted;
    }
    while (!pipe.empty()) retire();
    out.flush();
}

}  // namespace

int main(int argc, char **argv) {
    const options o = parse(argc, argv);
    try {
        if (o.decompress) {
            decompress_file(o);
        } else {
            compress_file(o);
        }
        return EXIT_SUCCESS;
    } catch (std::exception &e) {
        fprintf(stderr, "%s\n", e.what());
        return EXIT_FAILURE;
    }
}
