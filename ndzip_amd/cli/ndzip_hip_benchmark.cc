// ndzip-hip-benchmark -- the `ndzip-hip` rows of the reference's benchmark table.
//
// Reads the reference harness' dataset list (src/benchmark/benchmark.cc:102-125: one `file;float|double;n0 [n1 [n2]]` line per
// dataset, paths relative to the CSV) and writes its result CSV (benchmark.cc:1332-1337,1487-1489: same header, same columns,
// times as comma-separated integer microseconds), so the rows can be appended to a run of the reference harness and fed to its
// plot_benchmark.py unchanged.  Protocol of benchmark.cc:199-227,331-342: one warm-up, then repetitions until `--min-reps` and
// `--time-each` (and at most `--max-reps`) are satisfied; the recorded time is the offloader's kernel_duration (device pipeline by
// events, transfers excluded -- what the reference records for its GPU targets); the decompressed array must equal the input.
// Algorithm name: `ndzip-hip`, tunable 1, threads 1.
#include <cerrno>
#include <cinttypes>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/ndzip_hip.h"

namespace {

struct dataset {
    std::string path, name;
    int dtype;
    std::vector<uint32_t> extent;
};

void check(int status, const std::string &what) {
    if (status != NDZIP_HIP_OK) throw std::runtime_error(what + ": " + ndzip_hip_last_error());
}

std::vector<dataset> load_metadata_file(const std::string &csv) {
    FILE *f = fopen(csv.c_str(), "r");
    if (!f) throw std::runtime_error("fopen: " + csv + ": " + strerror(errno));
    const size_t slash = csv.find_last_of('/');
    const std::string dir = slash == std::string::npos ? "" : csv.substr(0, slash + 1);
    std::vector<dataset> out;
    char line[1024];
    while (fgets(line, sizeof line, f)) {
        line[strcspn(line, "\r\n")] = 0;  // (std::getline drops the newline)
        char name[100], type[10] = "";
        unsigned long long e[3];
        const int n = sscanf(line, "%99[^;];%9[^;];%llu %llu %llu", name, type, e, e + 1, e + 2);
        const bool is_float = strcmp(type, "float") == 0, is_double = strcmp(type, "double") == 0;
        if (n >= 3 && n <= 5 && (is_float || is_double)) {
            dataset d;
            d.path = dir + name;
            const char *base = strrchr(name, '/');
            d.name = base ? base + 1 : name;
            d.dtype = is_float ? NDZIP_HIP_F32 : NDZIP_HIP_F64;
            for (int i = 0; i < n - 2; ++i) {
                if (e[i] > 0xffffffffull) throw std::runtime_error(csv + ": extent too large: " + line);
                d.extent.push_back(static_cast<uint32_t>(e[i]));
            }
            out.push_back(d);
        } else if (n != 0) {  // as the reference: an empty line (sscanf -> EOF) is invalid too
            fclose(f);
            throw std::runtime_error(csv + ": Invalid line: " + line);
        }
    }
    fclose(f);
    return out;
}

struct params {
    uint64_t min_time_us = 1000 * 1000;
    unsigned min_reps = 1, max_reps = 100;
    bool warm_up = true;
};

// benchmark.cc:199-227
struct accumulator {
    std::vector<uint64_t> times;
    uint64_t total = 0;
    unsigned reps = 0;
    bool warmed_up = false;
    bool more(const params &p) const {
        const unsigned lo = p.min_reps > 1 ? p.min_reps : 1;
        return (p.warm_up && !warmed_up) || reps < lo || (total < p.min_time_us && reps < (p.max_reps > lo ? p.max_reps : lo));
    }
    void record(const params &p, uint64_t us) {
        if (p.warm_up && !warmed_up) {
            warmed_up = true;
        } else {
            times.push_back(us);
            total += us;
            ++reps;
        }
    }
};

void print_times(const std::vector<uint64_t> &t) {
    for (size_t i = 0; i < t.size(); ++i) printf("%s%" PRIu64, i ? "," : "", t[i]);
}

void benchmark_dataset(const dataset &d, const params &p) {
    const int dims = static_cast<int>(d.extent.size());
    const size_t wb = d.dtype == NDZIP_HIP_F32 ? 4 : 8;
    uint64_t n = 1, bound = 0;
    for (uint32_t e : d.extent) n *= e;
    check(ndzip_hip_compressed_length_bound(d.dtype, dims, d.extent.data(), &bound), d.name);
    void *input = nullptr, *stream = nullptr, *output = nullptr;
    ndzip_hip_offloader *off = nullptr;
    try {
        check(ndzip_hip_host_alloc(n * wb, &input), d.name);
        check(ndzip_hip_host_alloc(bound * wb, &stream), d.name);
        check(ndzip_hip_host_alloc(n * wb, &output), d.name);
        FILE *f = fopen(d.path.c_str(), "rb");
        if (!f) throw std::runtime_error("fopen: " + d.path + ": " + strerror(errno));
        const size_t got = fread(input, 1, n * wb, f);
        fclose(f);
        if (got != n * wb) throw std::runtime_error(d.path + ": file is shorter than its extent");
        check(ndzip_hip_offloader_create(d.dtype, dims, d.extent.data(), 1, &off), d.name);
        accumulator comp, decomp;
        uint32_t words = 0;
        while (comp.more(p)) {
            uint64_t ns = 0;
            check(ndzip_hip_offloader_submit_compress(off, 0, d.extent.data(), input, stream), d.name);
            check(ndzip_hip_offloader_wait(off, 0, &words, &ns), d.name);
            comp.record(p, ns / 1000);
        }
        while (decomp.more(p)) {
            uint64_t ns = 0;
            uint32_t consumed = 0;
            check(ndzip_hip_offloader_submit_decompress(off, 0, d.extent.data(), stream, words, output), d.name);
            check(ndzip_hip_offloader_wait(off, 0, &consumed, &ns), d.name);
            decomp.record(p, ns / 1000);
        }
        if (memcmp(input, output, n * wb) != 0) {
            throw std::logic_error("mismatch between input and decompressed buffer for " + d.name + " with ndzip-hip (tunable=1)");
        }
        printf("%s;%s;%d;ndzip-hip;1;1;", d.name.c_str(), d.dtype == NDZIP_HIP_F32 ? "float" : "double", dims);
        print_times(comp.times);
        printf(";");
        print_times(decomp.times);
        printf(";%" PRIu64 ";%" PRIu64 "\n", static_cast<uint64_t>(n * wb), static_cast<uint64_t>(words) * wb);
        fflush(stdout);
    } catch (...) {
        ndzip_hip_offloader_destroy(off);
        ndzip_hip_host_free(input);
        ndzip_hip_host_free(stream);
        ndzip_hip_host_free(output);
        throw;
    }
    ndzip_hip_offloader_destroy(off);
    ndzip_hip_host_free(input);
    ndzip_hip_host_free(stream);
    ndzip_hip_host_free(output);
}

[[noreturn]] void usage(const char *argv0, const std::string &msg) {
    fprintf(stderr,
            "%s\nUsage: %s [options] csv-file\n\n"
            "  --help               show this help\n"
            "  -t [ --time-each ] t repeat each for at least t ms (default 1000)\n"
            "  -r [ --min-reps ] n  repeat each at least n times (default 1)\n"
            "  -R [ --max-reps ] n  repeat each at most n times (default 100)\n"
            "  --no-warmup          do not perform an additional warm-up step per benchmark\n"
            "  -a [ --algorithms ]  accepted for compatibility; only ndzip-hip is available\n"
            "  --no-mmap            accepted, ignored\n",
            msg.c_str(), argv0);
    exit(msg.empty() ? EXIT_SUCCESS : EXIT_FAILURE);
}

}  // namespace

int main(int argc, char **argv) {
    params p;
    std::string csv;
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        auto num = [&](const char *name) -> unsigned long long {
            if (i + 1 >= argc) usage(argv[0], std::string("the required argument for option '") + name + "' is missing");
            char *end = nullptr;
            const unsigned long long v = strtoull(argv[++i], &end, 10);
            if (*end) usage(argv[0], std::string("the argument for option '") + name + "' is invalid");
            return v;
        };
        if (a == "--help") usage(argv[0], "");
        else if (a == "-t" || a == "--time-each") p.min_time_us = num("--time-each") * 1000;
        else if (a == "-r" || a == "--min-reps") p.min_reps = static_cast<unsigned>(num("--min-reps"));
        else if (a == "-R" || a == "--max-reps") p.max_reps = static_cast<unsigned>(num("--max-reps"));
        else if (a == "--no-warmup") p.warm_up = false;
        else if (a == "--no-mmap") {}
        else if (a == "-a" || a == "--algorithms") {
            if (i + 1 >= argc) usage(argv[0], "the required argument for option '--algorithms' is missing");
            if (std::string(argv[++i]) != "ndzip-hip") usage(argv[0], std::string("unknown algorithm ") + argv[i] + " (available: ndzip-hip)");
        } else if (!a.empty() && a[0] == '-') usage(argv[0], "unrecognised option '" + a + "'");
        else if (csv.empty()) csv = a;
        else usage(argv[0], "too many positional options have been specified on the command line");
    }
    if (csv.empty()) usage(argv[0], "the option '--csv-file' is required but missing");
    try {
        const std::vector<dataset> sets = load_metadata_file(csv);
        printf("dataset;data type;dimensions;algorithm;tunable;number of threads;"
               "compression times (microseconds);decompression times (microseconds);"
               "uncompressed bytes;compressed bytes\n");
        for (const dataset &d : sets) benchmark_dataset(d, p);
        return EXIT_SUCCESS;
    } catch (std::exception &e) {
        fprintf(stderr, "fatal: %s\n", e.what());
        return EXIT_FAILURE;
    }
}
