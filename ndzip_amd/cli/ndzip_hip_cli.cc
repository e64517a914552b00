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
//
// I/O follows src/io/io.cc, with one difference in the default: files, stdin and stdout go through stdio and pinned staging
// buffers (stdio_input_stream / stdio_output_stream, io.cc:17-116) unless `--mmap` is given; `--no-mmap` -- the reference's
// switch, whose default is the mapped path (compress.cc:150,211-216) -- is accepted and selects what already is the default.
// With `--mmap` regular files are memory-mapped (mmap_input_stream / mmap_output_stream, io.cc:118-256: the input is mapped once
// and handed to the device copy in place; the output file grows by one mapped window per array and is truncated to its final
// length).  The mapped path asks the HIP runtime to copy to and from file-backed mappings, which it has to stage page by page;
// it stays opt-in until it has been run and timed against the pinned path on the hardware.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
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
    bool use_mmap = false;  // --mmap: memory-mapped file I/O (the reference's default, compress.cc:150,211-216)
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
            "  --mmap                    memory-mapped I/O for regular files (stdin / stdout never are)\n"
            "  --no-mmap                 stdio and pinned staging buffers (the default)\n"
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
        } else if (a == "--mmap") {
            o.use_mmap = true;
        } else if (a == "--no-mmap") {
            o.use_mmap = false;
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

bool is_path(const std::string &name) { return !name.empty() && name != "-"; }

// ---- input: a run of bytes consumed front to back; `take` returns a pointer to the next `bytes` bytes (valid until the job
// that uses them has been retired), which is either inside the mapping or the caller's own buffer filled by fread ------------
struct input {
    virtual ~input() = default;
    // up to `bytes` bytes: *got = how many there were (0 at the end of the input)
    virtual const void *take(size_t bytes, void *own_buffer, size_t *got) = 0;
    // mapped input only: everything that is left, without consuming it
    virtual bool is_mapped() const { return false; }
    virtual const void *peek(size_t *available) {
        *available = 0;
        return nullptr;
    }
};

struct stdio_input final : input {  // io.cc:17-66
    FILE *f = stdin;
    bool owned = false;
    explicit stdio_input(const std::string &name) {
        if (is_path(name)) {
            f = fopen(name.c_str(), "rb");
            if (!f) throw std::runtime_error("fopen: " + name + ": " + strerror(errno));
            owned = true;
        }
    }
    ~stdio_input() override {
        if (owned) fclose(f);
    }
    const void *take(size_t bytes, void *own_buffer, size_t *got) override {
        *got = fread(own_buffer, 1, bytes, f);
        if (*got < bytes && ferror(f)) throw std::runtime_error(std::string("fread: ") + strerror(errno));
        return own_buffer;
    }
};

struct mapped_input final : input {  // io.cc:118-176
    int fd = -1;
    void *map = nullptr;
    size_t size = 0, offset = 0;
    explicit mapped_input(const std::string &name) {
        fd = open(name.c_str(), O_RDONLY);
        if (fd == -1) throw std::runtime_error("open: " + name + ": " + strerror(errno));
        struct stat st {};
        if (fstat(fd, &st) == -1) {
            close(fd);
            throw std::runtime_error("fstat: " + name + ": " + strerror(errno));
        }
        size = static_cast<size_t>(st.st_size);
        if (size) {
            // (a private, WRITABLE view although it is never written: the HIP runtime may pin the pages it copies from, and
            // pinning a read-only mapping can be refused)
            map = mmap(nullptr, size, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_POPULATE, fd, 0);
            if (map == MAP_FAILED) {
                close(fd);
                throw std::runtime_error("mmap: " + name + ": " + strerror(errno));
            }
        }
    }
    ~mapped_input() override {
        if (map) munmap(map, size);
        close(fd);
    }
    const void *take(size_t bytes, void *, size_t *got) override {
        const char *p = static_cast<const char *>(map) + offset;
        *got = bytes < size - offset ? bytes : size - offset;
        offset += *got;
        return p;
    }
    bool is_mapped() const override { return true; }
    const void *peek(size_t *available) override {
        *available = size - offset;
        return static_cast<const char *>(map) + offset;
    }
};

// ---- output: arrays are committed in order; `window` is where the next `max_bytes` bytes may be produced in place (nullptr:
// produce them in your own buffer and pass it to commit) ----------------------------------------------------------------------
struct output {
    virtual ~output() noexcept(false) {}
    virtual void *window(size_t max_bytes) = 0;
    virtual void commit(const void *data, size_t bytes) = 0;  // `data`: the window handed out for this array, or an own buffer
    virtual void finish() = 0;
};

struct stdio_output final : output {  // io.cc:69-116
    FILE *f = stdout;
    bool owned = false;
    explicit stdio_output(const std::string &name) {
        if (is_path(name)) {
            f = fopen(name.c_str(), "wb");
            if (!f) throw std::runtime_error("fopen: " + name + ": " + strerror(errno));
            owned = true;
        }
    }
    ~stdio_output() noexcept(false) override {
        if (owned) fclose(f);
    }
    void *window(size_t) override { return nullptr; }
    void commit(const void *data, size_t bytes) override {
        if (bytes && fwrite(data, bytes, 1, f) < 1) throw std::runtime_error(std::string("fwrite: ") + strerror(errno));
    }
    void finish() override {
        if (fflush(f) != 0) throw std::runtime_error(std::string("fflush: ") + strerror(errno));
    }
};

// The file grows by one mapped window per array (ftruncate + mmap of the page-aligned range that holds it) and is truncated to
// the committed length at the end -- mmap_output_stream (io.cc:178-256) with several windows alive at a time, because several
// arrays are in flight and are produced in place by concurrent device copies.  A window is handed out behind the window of the
// array before it (each as long as that array's MAXIMUM size: live windows must not overlap); an array that turns out shorter
// than its maximum leaves every later window too far back, so commit() copies such an array forward to its final place and
// gives the blocks of its provisional place back to the file system (FALLOC_FL_PUNCH_HOLE, best effort).  The file's LENGTH
// therefore reaches the sum of the maxima until finish() truncates it, but it is sparse: the blocks in use stay near the
// committed bytes plus the arrays in flight.
struct mapped_output final : output {
    struct win {
        void *base;      // mmap result
        size_t map_len;  // length of the mapping
        char *data;      // where the array was produced
        size_t offset;   // provisional file offset of `data`
    };
    int fd = -1;
    size_t committed = 0, reserved = 0, page;
    std::vector<win> live;  // in hand-out order
    explicit mapped_output(const std::string &name) : page(static_cast<size_t>(sysconf(_SC_PAGESIZE))) {
        fd = open(name.c_str(), O_RDWR | O_TRUNC | O_CREAT, static_cast<mode_t>(0666));
        if (fd == -1) throw std::runtime_error("open: " + name + ": " + strerror(errno));
    }
    ~mapped_output() noexcept(false) override {
        for (auto &w : live) munmap(w.base, w.map_len);
        close(fd);
    }
    void *window(size_t max_bytes) override {
        const size_t offset = reserved;
        reserved += max_bytes;
        if (ftruncate(fd, static_cast<off_t>(reserved)) == -1) throw std::runtime_error(std::string("ftruncate: ") + strerror(errno));
        const size_t aligned = offset / page * page, len = offset - aligned + (max_bytes ? max_bytes : 1);
        void *base = mmap(nullptr, len, PROT_READ | PROT_WRITE, MAP_SHARED, fd, static_cast<off_t>(aligned));
        if (base == MAP_FAILED) throw std::runtime_error(std::string("mmap: ") + strerror(errno));
        live.push_back({base, len, static_cast<char *>(base) + (offset - aligned), offset});
        return live.back().data;
    }
    void commit(const void *data, size_t bytes) override {
        if (live.empty() || live.front().data != data) throw std::runtime_error("output windows are committed in hand-out order");
        win w = live.front();
        live.erase(live.begin());
        const bool moved = w.offset != committed && bytes;
        if (moved) {  // an earlier array was shorter than its window: copy this one up to its place (forward copy, destination first)
            if (pwrite(fd, w.data, bytes, static_cast<off_t>(committed)) != static_cast<ssize_t>(bytes)) {
                throw std::runtime_error(std::string("pwrite: ") + strerror(errno));
            }
        }
        if (munmap(w.base, w.map_len) == -1) throw std::runtime_error(std::string("munmap: ") + strerror(errno));
        committed += bytes;
        if (moved) {
            // whole pages of the provisional place that lie behind everything committed so far
            const size_t from = (std::max(w.offset, committed) + page - 1) / page * page, to = (w.offset + bytes) / page * page;
            if (to > from) (void) fallocate(fd, FALLOC_FL_PUNCH_HOLE | FALLOC_FL_KEEP_SIZE, static_cast<off_t>(from), static_cast<off_t>(to - from));
        }
    }
    void finish() override {
        if (ftruncate(fd, static_cast<off_t>(committed)) == -1) throw std::runtime_error(std::string("ftruncate: ") + strerror(errno));
    }
};

std::unique_ptr<input> open_input(const options &o) {
    if (o.use_mmap && is_path(o.input)) return std::make_unique<mapped_input>(o.input);
    return std::make_unique<stdio_input>(o.input);
}

std::unique_ptr<output> open_output(const options &o) {
    if (o.use_mmap && is_path(o.output)) return std::make_unique<mapped_output>(o.output);
    return std::make_unique<stdio_output>(o.output);
}

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
    auto in = open_input(o);
    auto out = open_output(o);
    offloader off(o.dtype, z.dims, o.size.data(), o.slots);
    job_buffers buf(o.slots, z.chunk_bytes, z.bound_bytes);
    std::vector<void *> dest(o.slots, nullptr);  // where each slot's stream is produced: an output window or the pinned buffer
    pipeline pipe(o.slots);
    size_t compressed_words = 0;
    uint64_t kernel_ns = 0;
    const auto t0 = std::chrono::steady_clock::now();
    auto retire = [&] {
        const int slot = pipe.oldest_slot();
        uint32_t words = 0;
        uint64_t ns = 0;
        check(ndzip_hip_offloader_wait(off.h, slot, &words, &ns), "compress");
        out->commit(dest[slot], static_cast<size_t>(words) * z.wb);
        compressed_words += words;
        kernel_ns += ns;
        ++pipe.retired;
    };
    for (;;) {
        if (pipe.full()) retire();
        const int slot = pipe.next_slot();
        size_t got = 0;
        const void *chunk = in->take(z.chunk_bytes, buf.in[slot]->p, &got);
        if (got == 0) break;
        if (got != z.chunk_bytes) throw std::runtime_error("Input file size is not a multiple of the chunk size");  // io.cc read_exact
        void *w = out->window(z.bound_bytes);
        dest[slot] = w ? w : buf.out[slot]->p;
        check(ndzip_hip_offloader_submit_compress(off.h, slot, o.size.data(), chunk, dest[slot]), "compress");
        ++pipe.submitted;
    }
    while (!pipe.empty()) retire();
    out->finish();
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
    auto in = open_input(o);
    auto out = open_output(o);
    offloader off(o.dtype, z.dims, o.size.data(), o.slots);
    job_buffers buf(o.slots, z.bound_bytes, z.chunk_bytes);
    std::vector<void *> dest(o.slots, nullptr);
    pipeline pipe(o.slots);
    auto retire = [&] {
        const int slot = pipe.oldest_slot();
        check(ndzip_hip_offloader_wait(off.h, slot, nullptr, nullptr), "decompress");
        out->commit(dest[slot], z.chunk_bytes);
        ++pipe.retired;
    };
    for (;;) {
        if (pipe.full()) retire();
        const int slot = pipe.next_slot();
        // one stream = header (whose last entry gives the body length) + bodies + border
        const void *stream = nullptr;
        uint32_t words = 0;
        size_t available = 0;
        if (in->is_mapped()) {
            // mapped input: the stream is sized, validated and used in place
            const void *mapped = in->peek(&available);
            if (available == 0) break;
            check(ndzip_hip_stream_words(o.dtype, z.dims, o.size.data(), mapped, available / z.wb, &words), "stream header");
            size_t got = 0;
            stream = in->take(static_cast<size_t>(words) * z.wb, nullptr, &got);
        } else {
            // stdio: read the header, then the rest
            char *dst = static_cast<char *>(buf.in[slot]->p);
            size_t got = 0;
            in->take(z.header_bytes, dst, &got);
            if (got == 0 && z.header_bytes > 0) break;
            if (got != z.header_bytes) throw std::runtime_error("truncated stream header in input");
            check(ndzip_hip_stream_words(o.dtype, z.dims, o.size.data(), dst, z.bound_words, &words), "stream header");
            const size_t rest = static_cast<size_t>(words) * z.wb - z.header_bytes;
            size_t got_rest = 0;
            in->take(rest, dst + z.header_bytes, &got_rest);
            if (z.header_bytes == 0 && got_rest == 0) break;  // array without hypercubes: the stream is the border alone
            if (got_rest != rest) throw std::runtime_error("truncated stream in input");
            stream = dst;
        }
        void *w = out->window(z.chunk_bytes);
        dest[slot] = w ? w : buf.out[slot]->p;
        check(ndzip_hip_offloader_submit_decompress(off.h, slot, o.size.data(), stream, words, dest[slot]), "decompress");
        ++pipe.submitted;
    }
    while (!pipe.empty()) retire();
    out->finish();
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
