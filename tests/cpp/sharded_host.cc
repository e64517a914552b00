// A C++ host of the multi-GPU path over include/ndzip_hip_sharded.h -- the example INTEGRATION.md shows, and the program
// tests/test_hip_sharded_native.py runs (one process per GPU, RCCL over xGMI for the two exchanges and nothing else).
//
//   sharded_host --rank R --world N --id-file PATH --dtype f32|f64 --extent a[,b[,c]] --in ARRAY.bin --out STREAM.bin [--device D]
//
// Every rank reads its slab of the raw row-major array, compresses it on its GPU, takes part in the exchange, and copies its
// pieces to their place in ONE shared mapping of the output file: the result is the reference's single stream for the whole
// array, bit for bit (include/ndzip/ndzip.hh:227-240 on the global array).  Then the way back: every rank takes its pieces out of
// that file, decodes its slab without any collective and compares it with what it read.
// The ncclUniqueId travels through a file (rank 0 writes, the others poll): any transport a real application has will do.
//
// Build (one line):
//   g++ -std=c++17 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude tests/cpp/sharded_host.cc -o sharded_host
//       -Lndzip_amd -lndzip_hip_rccl -lndzip_hip -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,$PWD/ndzip_amd
#include <fcntl.h>
#include <hip/hip_runtime_api.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "ndzip_hip_sharded.h"

namespace {

[[noreturn]] void die(const char *what, const char *detail = "") {
    fprintf(stderr, "sharded_host: %s%s%s\n", what, *detail ? ": " : "", detail);
    exit(1);
}

void ok(int status, const char *what) {
    if (status != NDZIP_HIP_OK) die(what, ndzip_hip_sharded_last_error());
}

void hip_ok(hipError_t e, const char *what) {
    if (e != hipSuccess) die(what, hipGetErrorString(e));
}

bool wait_for_file(const std::string &path, size_t bytes, int seconds) {
    for (int i = 0; i < seconds * 20; ++i) {
        struct stat st;
        if (stat(path.c_str(), &st) == 0 && static_cast<size_t>(st.st_size) >= bytes) return true;
        std::this_thread::sleep_for(std::chrono::milliseconds(50));
    }
    return false;
}

}  // namespace

int main(int argc, char **argv) {
    uint32_t rank = 0, world = 1, extent[3] = {0, 0, 0};
    int dims = 0, dtype = NDZIP_HIP_F32, device = -1;
    std::string id_file, in_file, out_file;
    for (int i = 1; i + 1 < argc; i += 2) {
        const std::string k = argv[i], v = argv[i + 1];
        if (k == "--rank") rank = static_cast<uint32_t>(atoi(v.c_str()));
        else if (k == "--world") world = static_cast<uint32_t>(atoi(v.c_str()));
        else if (k == "--device") device = atoi(v.c_str());
        else if (k == "--id-file") id_file = v;
        else if (k == "--in") in_file = v;
        else if (k == "--out") out_file = v;
        else if (k == "--dtype") dtype = v == "f64" ? NDZIP_HIP_F64 : NDZIP_HIP_F32;
        else if (k == "--extent") {
            for (const char *p = v.c_str(); *p && dims < 3;) {
                extent[dims++] = static_cast<uint32_t>(strtoul(p, const_cast<char **>(&p), 10));
                if (*p == ',') ++p;
            }
        } else die("unknown option", k.c_str());
    }
    if (dims == 0 || in_file.empty() || out_file.empty() || (world > 1 && id_file.empty())) die("usage: see the head of tests/cpp/sharded_host.cc");
    const size_t wb = dtype == NDZIP_HIP_F32 ? 4 : 8;

    hip_ok(hipSetDevice(device >= 0 ? device : static_cast<int>(rank)), "hipSetDevice");
    hipStream_t stream;
    hip_ok(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking), "hipStreamCreate");

    // ---- the communicator: 128 bytes from rank 0 to everybody, by file ---------------------------------------------------
    void *comm = nullptr;
    if (world > 1) {
        char id[NDZIP_HIP_RCCL_UNIQUE_ID_BYTES];
        if (rank == 0) {
            ok(ndzip_hip_rccl_unique_id(id), "ncclGetUniqueId");
            const std::string tmp = id_file + ".tmp";
            FILE *f = fopen(tmp.c_str(), "wb");
            if (!f || fwrite(id, 1, sizeof id, f) != sizeof id || fclose(f) != 0) die("writing the id file", tmp.c_str());
            if (rename(tmp.c_str(), id_file.c_str()) != 0) die("publishing the id file", id_file.c_str());
        } else {
            if (!wait_for_file(id_file, sizeof id, 60)) die("rank 0 never published the id file", id_file.c_str());
            FILE *f = fopen(id_file.c_str(), "rb");
            if (!f || fread(id, 1, sizeof id, f) != sizeof id) die("reading the id file", id_file.c_str());
            fclose(f);
        }
        ok(ndzip_hip_rccl_comm_create(id, rank, world, &comm), "ncclCommInitRank");
    }

    // ---- plan, slab in ----------------------------------------------------------------------------------------------------
    ndzip_hip_sharded *codec = nullptr;
    ok(ndzip_hip_sharded_create(dtype, dims, extent, rank, world, comm, stream, &codec), "ndzip_hip_sharded_create");
    ndzip_hip_shard sh;
    ok(ndzip_hip_sharded_shard(codec, &sh), "shard");
    size_t row = wb, slab_elems = sh.extent[0];
    for (int d = 1; d < dims; ++d) {
        row *= extent[d];
        slab_elems *= sh.extent[d];
    }
    std::vector<char> slab(slab_elems * wb), back(slab_elems * wb);
    {
        FILE *f = fopen(in_file.c_str(), "rb");
        if (!f || fseek(f, static_cast<long>(sh.start0 * row), SEEK_SET) != 0 || fread(slab.data(), 1, slab.size(), f) != slab.size()) die("reading the slab", in_file.c_str());
        fclose(f);
    }
    void *d_in = nullptr, *d_out = nullptr;
    hip_ok(hipMalloc(&d_in, slab.size() ? slab.size() : 1), "hipMalloc");
    hip_ok(hipMalloc(&d_out, slab.size() ? slab.size() : 1), "hipMalloc");
    hip_ok(hipMemcpyAsync(d_in, slab.data(), slab.size(), hipMemcpyHostToDevice, stream), "H2D");

    // ---- compress: codec launch, two collectives; everything stays on the stream ----------------------------------------
    ok(ndzip_hip_sharded_compress(codec, d_in), "compress");
    ok(ndzip_hip_sharded_check(codec), "check after compress");

    // ---- this rank's pieces into the one stream ---------------------------------------------------------------------------
    ndzip_hip_stream_layout lay;
    ok(ndzip_hip_sharded_stream_layout(codec, &lay), "stream_layout");
    const size_t stream_bytes = lay.stream_words * wb;
    const int fd = open(out_file.c_str(), O_RDWR | O_CREAT, 0644);
    if (fd < 0 || ftruncate(fd, static_cast<off_t>(stream_bytes)) != 0) die("creating the output file", out_file.c_str());  // (every rank: same size)
    void *map = stream_bytes ? mmap(nullptr, stream_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0) : nullptr;
    if (stream_bytes && map == MAP_FAILED) die("mapping the output file", out_file.c_str());
    ok(ndzip_hip_sharded_write_stream(codec, map, lay.stream_words, rank == 0), "write_stream");
    if (stream_bytes) msync(map, stream_bytes, MS_SYNC);

    // ---- decode the resident body (no collective) ... ---------------------------------------------------------------------
    ok(ndzip_hip_sharded_decompress(codec, d_out), "decompress");
    ok(ndzip_hip_sharded_check(codec), "check after decompress");
    hip_ok(hipMemcpyAsync(back.data(), d_out, back.size(), hipMemcpyDeviceToHost, stream), "D2H");
    hip_ok(hipStreamSynchronize(stream), "sync");
    if (memcmp(back.data(), slab.data(), slab.size()) != 0) die("round trip of the resident stream differs from the slab");

    // ---- ... and, on a second handle, from the FILE once every rank has written its pieces --------------------------------
    // (the exchange doubles as the barrier: a compress on the same communicator returns its lengths only when all ranks are here)
    ok(ndzip_hip_sharded_compress(codec, d_in), "compress (barrier)");
    ok(ndzip_hip_sharded_check(codec), "check (barrier)");
    ndzip_hip_sharded *reader = nullptr;
    ok(ndzip_hip_sharded_create(dtype, dims, extent, rank, world, comm, stream, &reader), "create (reader)");
    ok(ndzip_hip_sharded_load(reader, map, lay.stream_words), "load");
    hip_ok(hipMemsetAsync(d_out, 0xff, back.size() ? back.size() : 1, stream), "memset");
    ok(ndzip_hip_sharded_decompress(reader, d_out), "decompress (from the file)");
    ok(ndzip_hip_sharded_check(reader), "check (reader)");
    hip_ok(hipMemcpyAsync(back.data(), d_out, back.size(), hipMemcpyDeviceToHost, stream), "D2H");
    hip_ok(hipStreamSynchronize(stream), "sync");
    if (memcmp(back.data(), slab.data(), slab.size()) != 0) die("round trip through the stream file differs from the slab");

    printf("rank %u/%u: slab rows [%u, %u), hypercubes [%u, %u), runs %llu words at %llu, border %llu words at %llu, stream %llu words: ok\n", rank, world,
            sh.start0, sh.start0 + sh.extent[0], sh.hc_begin, sh.hc_end, static_cast<unsigned long long>(lay.runs_words),
            static_cast<unsigned long long>(lay.runs_offset_words), static_cast<unsigned long long>(lay.border_words),
            static_cast<unsigned long long>(lay.border_offset_words), static_cast<unsigned long long>(lay.stream_words));
    if (stream_bytes) munmap(map, stream_bytes);
    close(fd);
    ok(ndzip_hip_sharded_destroy(reader), "destroy");
    ok(ndzip_hip_sharded_destroy(codec), "destroy");
    ok(ndzip_hip_rccl_comm_destroy(comm), "ncclCommDestroy");
    (void) hipFree(d_in);
    (void) hipFree(d_out);
    (void) hipStreamDestroy(stream);
    return 0;
}
