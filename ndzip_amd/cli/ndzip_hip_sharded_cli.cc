// ndzip-hip-sharded -- ONE array, ONE ndzip stream, N GPUs: the file-level tool of the multi-GPU path (include/ndzip_hip_sharded.h).
//
// The reference's `compress` tool (src/compress/compress.cc) drives one device; larger inputs are cut into independent arrays of `-n`
// elements (compress.cc:34-45).  This tool keeps the file format of ONE array -- the stream it writes is byte for byte what the
// reference (any back-end) writes for the whole array, so `compress -d` / `ndzip-hip -d` read it back, and the other way round --
// and spreads the work: dimension 0 is cut into one slab of whole hypercube planes per rank, every rank compresses its slab on its
// GPU, the only exchange is the all-gather of one length per rank and of the header segments, and every rank copies its pieces to
// their place in a shared mapping of the output file.  Decompression needs no exchange at all: a rank takes its pieces out of the
// mapped stream (validated header first) and writes its slab of the output.
//
//   ndzip-hip-sharded    -n 2048 1024 1024 -t float -i field.f32 -o field.ndz      # == compress -n 2048 1024 1024 ... of the reference
//   ndzip-hip-sharded -d -n 2048 1024 1024 -t float -i field.ndz -o field.f32
//
// Ranks are threads of this process, rank r on device r % devices.  --exchange rccl (default when there is one rank per device and
// more than one): each thread brings up its ncclComm_t through the library's bootstrap helpers and the two all-gathers are RCCL over
// xGMI; --exchange local (default otherwise -- several ranks per GPU, or a build of the library without RCCL): the library's in-process
// group, a rendezvous + device-to-device copies.  Options as in the reference tool where they exist (compress.cc:135-211):
// -d, -n, -t, -i, -o; plus --ranks N, --devices D, --exchange rccl|local, --repeat R (time R compress or decompress passes), -q.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cinttypes>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/ndzip_hip_sharded.h"

// (the RCCL transport is a separate translation unit of the library: a build without it -- the CPU suite's, against the kernels'
// functional model -- still links this tool, which then offers the local exchange only)
extern "C" {
int ndzip_hip_sharded_create(int, int, const uint32_t *, uint32_t, uint32_t, void *, void *, ndzip_hip_sharded **) __attribute__((weak));
int ndzip_hip_rccl_unique_id(void *) __attribute__((weak));
int ndzip_hip_rccl_comm_create(const void *, uint32_t, uint32_t, void **) __attribute__((weak));
int ndzip_hip_rccl_comm_destroy(void *) __attribute__((weak));
}

namespace {

struct options {
    bool decompress = false, quiet = false;
    std::vector<uint32_t> size;
    int dtype = NDZIP_HIP_F32;
    std::string input, output, exchange;
    int ranks = 0, devices = 0, repeat = 1;
};

[[noreturn]] void usage_error(const std::string &msg, const char *argv0) {
    if (!msg.empty()) fprintf(stderr, "%s\n\n", msg.c_str());
    fprintf(stderr,
            "Usage: %s [options]\n\nCompress or decompress ONE binary float dump as ONE ndzip stream on several GPUs:\n"
            "  --help                    show this help\n"
            "  -d [ --decompress ]       decompress (default compress)\n"
            "  -n [ --array-size ] arg   array size (one value per dimension, first-major)\n"
            "  -t [ --data-type ] arg    float|double (default float)\n"
            "  -i [ --input ] arg        input file\n"
            "  -o [ --output ] arg       output file\n"
            "  --ranks arg               ranks = slabs of dimension 0 (default: one per visible GPU)\n"
            "  --devices arg             GPUs to use (default: all visible); rank r runs on device r %% devices\n"
            "  --exchange arg            rccl|local (default: rccl with one rank per GPU, local otherwise)\n"
            "  --repeat arg              run the device part arg times and report the rate (default 1)\n"
            "  -q [ --quiet ]            no summary line\n",
            argv0);
    exit(msg.empty() ? 0 : 2);
}

options parse(int argc, char **argv) {
    options o;
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        auto value = [&](const char *name) -> std::string {
            if (i + 1 >= argc) usage_error(std::string("the required argument for option '") + name + "' is missing", argv[0]);
            return argv[++i];
        };
        if (a == "--help") usage_error("", argv[0]);
        else if (a == "-d" || a == "--decompress") o.decompress = true;
        else if (a == "-q" || a == "--quiet") o.quiet = true;
        else if (a == "-n" || a == "--array-size") {
            while (i + 1 < argc && argv[i + 1][0] != '-') {
                char *end = nullptr;
                const unsigned long long v = strtoull(argv[++i], &end, 10);
                if (*end || v > 0xffffffffull) usage_error(std::string("the argument ('") + argv[i] + "') for option '--array-size' is invalid", argv[0]);
                o.size.push_back(static_cast<uint32_t>(v));
            }
        } else if (a == "-t" || a == "--data-type") {
            const std::string t = value("--data-type");
            if (t == "float") o.dtype = NDZIP_HIP_F32;
            else if (t == "double") o.dtype = NDZIP_HIP_F64;
            else usage_error("Invalid data type " + t, argv[0]);  // compress.cc:203
        } else if (a == "-i" || a == "--input") o.input = value("--input");
        else if (a == "-o" || a == "--output") o.output = value("--output");
        else if (a == "--ranks") o.ranks = atoi(value("--ranks").c_str());
        else if (a == "--devices") o.devices = atoi(value("--devices").c_str());
        else if (a == "--repeat") o.repeat = atoi(value("--repeat").c_str());
        else if (a == "--exchange") {
            o.exchange = value("--exchange");
            if (o.exchange != "rccl" && o.exchange != "local") usage_error("Invalid exchange " + o.exchange, argv[0]);
        } else usage_error("unrecognised option '" + a + "'", argv[0]);
    }
    if (o.size.empty()) usage_error("the option '--array-size' is required but missing", argv[0]);
    if (o.size.size() > 3) usage_error("Expected between 1 and 3 dimensions, got " + std::to_string(o.size.size()), argv[0]);  // compress.cc:191-193
    if (o.input.empty() || o.output.empty()) usage_error("the options '--input' and '--output' are required (one array, mapped files)", argv[0]);
    if (o.ranks < 0 || o.devices < 0 || o.repeat < 1) usage_error("--ranks / --devices / --repeat must be positive", argv[0]);
    return o;
}

std::mutex report;
std::atomic<int> failures{0};

[[noreturn]] void die(uint32_t rank, const char *what, const char *detail) {
    {
        std::lock_guard<std::mutex> l(report);
        fprintf(stderr, "ndzip-hip-sharded: rank %u: %s: %s\n", rank, what, detail);
    }
    _exit(1);  // (the other ranks would wait for this one at the next rendezvous for ever)
}

void ok(uint32_t rank, int status, const char *what) {
    if (status != NDZIP_HIP_OK) die(rank, what, ndzip_hip_sharded_last_error());
}

struct mapping {
    void *p = nullptr;
    size_t bytes = 0;
    int fd = -1;
    void open_read(const std::string &path) {
        fd = ::open(path.c_str(), O_RDONLY);
        struct stat st;
        if (fd < 0 || fstat(fd, &st) != 0) die(0, "cannot open the input", path.c_str());
        bytes = static_cast<size_t>(st.st_size);
        p = bytes ? mmap(nullptr, bytes, PROT_READ, MAP_PRIVATE, fd, 0) : nullptr;
        if (bytes && p == MAP_FAILED) die(0, "cannot map the input", path.c_str());
    }
    void create(const std::string &path, size_t n) {
        fd = ::open(path.c_str(), O_RDWR | O_CREAT | O_TRUNC, 0644);
        if (fd < 0 || ftruncate(fd, static_cast<off_t>(n)) != 0) die(0, "cannot create the output", path.c_str());
        bytes = n;
        p = n ? mmap(nullptr, n, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0) : nullptr;
        if (n && p == MAP_FAILED) die(0, "cannot map the output", path.c_str());
    }
    void close() {
        if (p && bytes) {
            msync(p, bytes, MS_SYNC);
            munmap(p, bytes);
        }
        if (fd >= 0) ::close(fd);
        p = nullptr;
        fd = -1;
    }
};

struct shared_state {
    options o;
    int dims = 0;
    uint32_t extent[3] = {0, 0, 0};
    uint32_t world = 1;
    int devices = 1;
    bool rccl = false;
    size_t wb = 4, array_bytes = 0;
    mapping in, out;
    ndzip_hip_local_group *group = nullptr;  // also the tool's own barrier
    // Ranks that SHARE a GPU take turns with their device work: the compress kernel is a persistent grid sized for the whole
    // device, and two of them side by side is the one workload that has hung a box (two processes on one GPU, round 1).  One mutex
    // per device, never held across a rendezvous.
    std::vector<std::mutex> device_turn;
    bool shared_devices = false;
    char nccl_id[NDZIP_HIP_RCCL_UNIQUE_ID_BYTES];
    std::vector<double> seconds;  // per rank: device part of the timed passes
    uint64_t stream_words = 0;
};

void rank_main(shared_state *S, uint32_t rank) {
    const options &o = S->o;
    ok(rank, ndzip_hip_sharded_set_device(static_cast<int>(rank) % S->devices), "selecting the device");
    ndzip_hip_sharded *codec = nullptr;
    void *comm = nullptr;
    if (S->rccl) {
        ok(rank, ndzip_hip_rccl_comm_create(S->nccl_id, rank, S->world, &comm), "ncclCommInitRank");
        ok(rank, ndzip_hip_sharded_create(o.dtype, S->dims, S->extent, rank, S->world, comm, nullptr, &codec), "creating the codec");
    } else {
        ok(rank, ndzip_hip_sharded_create_local(o.dtype, S->dims, S->extent, rank, S->world, S->group, nullptr, &codec), "creating the codec");
    }
    ndzip_hip_shard sh;
    ok(rank, ndzip_hip_sharded_shard(codec, &sh), "shard");
    size_t row = S->wb;
    for (int d = 1; d < S->dims; ++d) row *= S->extent[d];
    const size_t slab_offset = sh.start0 * row;

    std::mutex &turn = S->device_turn[rank % static_cast<uint32_t>(S->devices)];
    // one pass of the device part of compress: codec launch (alone on its device when the device is shared), then the exchange
    auto compress_once = [&](const char *slab) {
        if (S->shared_devices) {
            std::lock_guard<std::mutex> l(turn);
            ok(rank, ndzip_hip_sharded_compress_local_host(codec, slab), "compress");
            ok(rank, ndzip_hip_sharded_check(codec), "check");  // (drains the stream before the next rank's turn)
        } else {
            ok(rank, ndzip_hip_sharded_compress_local_host(codec, slab), "compress");
        }
        ok(rank, ndzip_hip_sharded_exchange(codec), "exchange");
    };
    if (!o.decompress) {
        const char *slab = static_cast<const char *>(S->in.p) + slab_offset;
        compress_once(slab);  // (warm-up and the pass whose stream is written)
        ok(rank, ndzip_hip_sharded_check(codec), "check");
        if (o.repeat > 1) {
            ndzip_hip_local_group_barrier(S->group);
            const auto t0 = std::chrono::steady_clock::now();
            for (int k = 0; k < o.repeat; ++k) compress_once(slab);
            ok(rank, ndzip_hip_sharded_check(codec), "check");
            S->seconds[rank] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        }
        ndzip_hip_stream_layout lay;
        ok(rank, ndzip_hip_sharded_stream_layout(codec, &lay), "stream_layout");
        if (rank == 0) {
            S->stream_words = lay.stream_words;
            S->out.create(o.output, lay.stream_words * S->wb);
        }
        ndzip_hip_local_group_barrier(S->group);  // the output is mapped
        ok(rank, ndzip_hip_sharded_write_stream(codec, S->out.p, lay.stream_words, rank == 0), "write_stream");
    } else {
        ok(rank, ndzip_hip_sharded_load(codec, S->in.p, S->in.bytes / S->wb), "load (is this a stream of an array of this size and type?)");
        char *slab = static_cast<char *>(S->out.p) + slab_offset;
        ndzip_hip_local_group_barrier(S->group);
        const auto t0 = std::chrono::steady_clock::now();
        for (int k = 0; k < o.repeat; ++k) {
            if (S->shared_devices) {
                std::lock_guard<std::mutex> l(turn);
                ok(rank, ndzip_hip_sharded_decompress_host(codec, slab), "decompress");  // (returns when the slab is on the host)
            } else {
                ok(rank, ndzip_hip_sharded_decompress_host(codec, slab), "decompress");
            }
        }
        ok(rank, ndzip_hip_sharded_check(codec), "check");
        S->seconds[rank] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
    ndzip_hip_local_group_barrier(S->group);  // every rank's pieces are in the output
    ok(rank, ndzip_hip_sharded_destroy(codec), "destroy");
    if (comm) ok(rank, ndzip_hip_rccl_comm_destroy(comm), "ncclCommDestroy");
}

}  // namespace

int main(int argc, char **argv) {
    shared_state S;
    S.o = parse(argc, argv);
    const options &o = S.o;
    S.dims = static_cast<int>(o.size.size());
    S.wb = o.dtype == NDZIP_HIP_F32 ? 4 : 8;
    uint64_t elements = 1;
    for (int d = 0; d < S.dims; ++d) {
        S.extent[d] = o.size[d];
        elements *= o.size[d];
    }
    S.array_bytes = elements * S.wb;
    int visible = 0;
    if (ndzip_hip_sharded_device_count(&visible) != NDZIP_HIP_OK) {
        fprintf(stderr, "ndzip-hip-sharded: %s\n", ndzip_hip_sharded_last_error());
        return 1;
    }
    S.devices = o.devices ? o.devices : visible;
    if (S.devices > visible) {
        fprintf(stderr, "ndzip-hip-sharded: --devices %d, but %d GPU(s) are visible\n", S.devices, visible);
        return 2;
    }
    S.world = static_cast<uint32_t>(o.ranks ? o.ranks : S.devices);
    const bool have_rccl = ndzip_hip_sharded_create && ndzip_hip_rccl_unique_id && ndzip_hip_rccl_comm_create && ndzip_hip_rccl_comm_destroy;
    const bool one_rank_per_device = S.world > 1 && static_cast<int>(S.world) <= S.devices;
    if (o.exchange == "rccl" && !have_rccl) {
        fprintf(stderr, "ndzip-hip-sharded: this build of the library has no RCCL transport\n");
        return 2;
    }
    if (o.exchange == "rccl" && !one_rank_per_device && S.world > 1) {
        fprintf(stderr, "ndzip-hip-sharded: --exchange rccl needs one rank per GPU (%u ranks, %d GPUs)\n", S.world, S.devices);
        return 2;
    }
    S.rccl = S.world > 1 && (o.exchange == "rccl" || (o.exchange.empty() && have_rccl && one_rank_per_device));

    // plan check on the host before anything is mapped or allocated: the shard of rank 0 is computed by the same code the ranks run
    ndzip_hip_shard probe;
    if (ndzip_hip_sharded_plan(o.dtype, S.dims, S.extent, 0, S.world, &probe) != NDZIP_HIP_OK) {
        fprintf(stderr, "ndzip-hip-sharded: %s\n", ndzip_hip_sharded_last_error());
        return 2;
    }
    S.in.open_read(o.input);
    if (!o.decompress && S.in.bytes != S.array_bytes) {
        fprintf(stderr, "ndzip-hip-sharded: the input has %zu bytes, an array of this size and type %zu (one array per file)\n", S.in.bytes, S.array_bytes);
        return 2;
    }
    if (o.decompress) {
        if (S.in.bytes % S.wb != 0) {
            fprintf(stderr, "ndzip-hip-sharded: the input is not a whole number of stream words\n");
            return 2;
        }
        S.out.create(o.output, S.array_bytes);
    }
    if (ndzip_hip_local_group_create(S.world, &S.group) != NDZIP_HIP_OK) return 1;
    if (S.rccl && ndzip_hip_rccl_unique_id(S.nccl_id) != NDZIP_HIP_OK) {
        fprintf(stderr, "ndzip-hip-sharded: %s\n", ndzip_hip_sharded_last_error());
        return 1;
    }
    S.seconds.assign(S.world, 0.0);
    S.device_turn = std::vector<std::mutex>(static_cast<size_t>(S.devices));
    S.shared_devices = static_cast<int>(S.world) > S.devices;

    std::vector<std::thread> ranks;
    for (uint32_t r = 0; r < S.world; ++r) ranks.emplace_back(rank_main, &S, r);
    for (auto &t : ranks) t.join();
    const size_t out_bytes = S.out.bytes;
    S.out.close();
    S.in.close();
    ndzip_hip_local_group_destroy(S.group);

    if (!o.quiet) {
        const size_t compressed = o.decompress ? S.in.bytes : out_bytes;
        double slowest = 0;
        for (double s : S.seconds) slowest = s > slowest ? s : slowest;
        fprintf(stderr, "raw = %zu bytes, compressed = %zu bytes, ratio = %.4f, %u rank(s) on %d GPU(s), exchange %s", S.array_bytes, compressed,
                S.array_bytes ? static_cast<double>(compressed) / static_cast<double>(S.array_bytes) : 0.0, S.world, S.devices,
                S.world == 1 ? "none" : S.rccl ? "rccl" : "local");
        if (slowest > 0) {
            fprintf(stderr, ", %s x%d incl. host copies: %.3f GB/s", o.decompress ? "decompress" : "compress", o.repeat,
                    static_cast<double>(S.array_bytes) * o.repeat / slowest / 1e9);
        }
        fprintf(stderr, "\n");
    }
    return failures.load() ? 1 : 0;
}
