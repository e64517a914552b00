// Every rank of a sharded plan as a THREAD of one process, over include/ndzip_hip_sharded.h with a caller-supplied exchange
// (ndzip_hip_sharded_create_with_collectives): the all-gather is a rendezvous of the rank threads followed by device-to-device copies.
// Purpose: the world > 1 path of the C++ host -- length gather, offset kernel, (padded) header gather, compaction of unequal
// segments, stream layout, the way back through load -- on a box with ONE GPU (tests/test_hip_sharded_native.py) and, compiled against
// the kernels' functional model, in the CPU suite (tests/test_sharded_native_cpu.py) without any Python in the loop.
// A production host runs one process (or thread) per GPU over RCCL: tests/cpp/sharded_host.cc.  Here all ranks share device 0, so the
// persistent compress kernels of different ranks are kept from overlapping (one rank's device work at a time: `device_turn`).
//
//   sharded_threads --world N --dtype f32|f64 --extent a[,b[,c]] --in ARRAY.bin --out STREAM.bin
#include <hip/hip_runtime.h>

#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "ndzip_hip_sharded.h"

namespace {

struct rendezvous {
    std::mutex m;
    std::condition_variable cv;
    uint32_t world = 1, arrived = 0, generation = 0;
    std::vector<const uint32_t *> send;
    void wait() {  // a reusable barrier
        std::unique_lock<std::mutex> l(m);
        const uint32_t g = generation;
        if (++arrived == world) {
            arrived = 0;
            ++generation;
            cv.notify_all();
        } else {
            cv.wait(l, [&] { return generation != g; });
        }
    }
};

struct rank_ctx {
    rendezvous *rv;
    uint32_t rank;
};

std::mutex device_turn;  // one rank's kernels in flight at a time (all ranks share one device here)
std::mutex print_turn;
int failures = 0;

// ndzip_hip_collectives::all_gather_u32: stream-ordered on the CALLER's stream as the contract says -- the send buffer is complete when
// the stream has drained, every rank then copies every rank's segment on its own stream
int all_gather_u32(void *ctx, const uint32_t *d_send, uint32_t *d_recv, size_t count, void *hip_stream) {
    auto *c = static_cast<rank_ctx *>(ctx);
    auto stream = static_cast<hipStream_t>(hip_stream);
    if (hipStreamSynchronize(stream) != hipSuccess) return 1;
    c->rv->send[c->rank] = d_send;
    c->rv->wait();
    for (uint32_t r = 0; r < c->rv->world; ++r) {
        if (hipMemcpyAsync(d_recv + r * count, c->rv->send[r], count * sizeof(uint32_t), hipMemcpyDeviceToDevice, stream) != hipSuccess) return 2;
    }
    if (hipStreamSynchronize(stream) != hipSuccess) return 3;
    c->rv->wait();  // nobody overwrites its send buffer before everybody has read it
    return 0;
}

const char *error_string(void *, int code) { return code == 1 ? "stream did not drain" : code == 2 ? "device copy failed" : "copies did not complete"; }

#define CHECK(expr, what)                                                                                      \
    do {                                                                                                       \
        if ((expr) != NDZIP_HIP_OK) {                                                                          \
            std::lock_guard<std::mutex> l(print_turn);                                                         \
            fprintf(stderr, "rank %u: %s: %s\n", rank, what, ndzip_hip_sharded_last_error());                  \
            exit(1); /* (the other ranks would wait for this one at the next rendezvous for ever) */          \
        }                                                                                                      \
    } while (0)

void run_rank(uint32_t rank, uint32_t world, int dtype, int dims, const uint32_t *extent, const std::vector<char> *array, std::vector<char> *stream_out,
        rendezvous *rv) {
    const size_t wb = dtype == NDZIP_HIP_F32 ? 4 : 8;
    rank_ctx ctx{rv, rank};
    const ndzip_hip_collectives table{&ctx, all_gather_u32, error_string};
    hipStream_t stream = nullptr;
    if (hipStreamCreateWithFlags(&stream, hipStreamNonBlocking) != hipSuccess) {
        ++failures;
        return;
    }
    ndzip_hip_sharded *codec = nullptr, *reader = nullptr;
    {
        std::lock_guard<std::mutex> l(device_turn);
        CHECK(ndzip_hip_sharded_create_with_collectives(dtype, dims, extent, rank, world, &table, stream, &codec), "create");
        CHECK(ndzip_hip_sharded_create_with_collectives(dtype, dims, extent, rank, world, &table, stream, &reader), "create (reader)");
    }
    ndzip_hip_shard sh;
    CHECK(ndzip_hip_sharded_shard(codec, &sh), "shard");
    size_t row = wb, slab_bytes = static_cast<size_t>(sh.extent[0]) * wb;
    for (int d = 1; d < dims; ++d) {
        row *= extent[d];
        slab_bytes *= sh.extent[d];
    }
    const char *slab = array->data() + sh.start0 * row;
    std::vector<char> back(slab_bytes);
    void *d_in = nullptr, *d_out = nullptr;
    if (hipMalloc(&d_in, slab_bytes ? slab_bytes : 1) != hipSuccess || hipMalloc(&d_out, slab_bytes ? slab_bytes : 1) != hipSuccess) {
        ++failures;
        return;
    }
    (void) hipMemcpyAsync(d_in, slab, slab_bytes, hipMemcpyHostToDevice, stream);

    for (int round = 0; round < 2; ++round) {  // (twice: the handle is reused)
        {
            std::lock_guard<std::mutex> l(device_turn);
            CHECK(ndzip_hip_sharded_compress_local(codec, d_in), "compress_local");
            (void) hipStreamSynchronize(stream);
        }
        CHECK(ndzip_hip_sharded_exchange(codec), "exchange");  // (rendezvous inside: NOT under the device lock)
    }
    CHECK(ndzip_hip_sharded_check(codec), "check");
    ndzip_hip_stream_layout lay;
    CHECK(ndzip_hip_sharded_stream_layout(codec, &lay), "stream_layout");
    rv->wait();
    if (rank == 0) stream_out->assign(lay.stream_words * wb, 0);
    rv->wait();
    CHECK(ndzip_hip_sharded_write_stream(codec, stream_out->data(), lay.stream_words, rank == 0), "write_stream");
    {
        std::lock_guard<std::mutex> l(device_turn);
        CHECK(ndzip_hip_sharded_decompress(codec, d_out), "decompress");
        (void) hipMemcpyAsync(back.data(), d_out, slab_bytes, hipMemcpyDeviceToHost, stream);
        (void) hipStreamSynchronize(stream);
    }
    CHECK(ndzip_hip_sharded_check(codec), "check after decompress");
    bool ok = memcmp(back.data(), slab, slab_bytes) == 0;
    rv->wait();  // every rank's pieces are in the stream now
    {
        std::lock_guard<std::mutex> l(device_turn);
        CHECK(ndzip_hip_sharded_load(reader, stream_out->data(), lay.stream_words), "load");
        (void) hipMemsetAsync(d_out, 0xff, slab_bytes ? slab_bytes : 1, stream);
        CHECK(ndzip_hip_sharded_decompress(reader, d_out), "decompress (loaded)");
        (void) hipMemcpyAsync(back.data(), d_out, slab_bytes, hipMemcpyDeviceToHost, stream);
        (void) hipStreamSynchronize(stream);
    }
    CHECK(ndzip_hip_sharded_check(reader), "check (reader)");
    ok = ok && memcmp(back.data(), slab, slab_bytes) == 0;
    {
        std::lock_guard<std::mutex> l(print_turn);
        printf("rank %u/%u: hypercubes [%u, %u), runs %llu words at %llu, border %llu words at %llu: %s\n", rank, world, sh.hc_begin, sh.hc_end,
                static_cast<unsigned long long>(lay.runs_words), static_cast<unsigned long long>(lay.runs_offset_words),
                static_cast<unsigned long long>(lay.border_words), static_cast<unsigned long long>(lay.border_offset_words), ok ? "ok" : "ROUND TRIP DIFFERS");
        if (!ok) ++failures;
    }
    {
        std::lock_guard<std::mutex> l(device_turn);
        (void) ndzip_hip_sharded_destroy(reader);
        (void) ndzip_hip_sharded_destroy(codec);
        (void) hipFree(d_in);
        (void) hipFree(d_out);
    }
}

}  // namespace

int main(int argc, char **argv) {
    uint32_t world = 2, extent[3] = {0, 0, 0};
    int dims = 0, dtype = NDZIP_HIP_F32;
    std::string in_file, out_file;
    for (int i = 1; i + 1 < argc; i += 2) {
        const std::string k = argv[i], v = argv[i + 1];
        if (k == "--world") world = static_cast<uint32_t>(atoi(v.c_str()));
        else if (k == "--in") in_file = v;
        else if (k == "--out") out_file = v;
        else if (k == "--dtype") dtype = v == "f64" ? NDZIP_HIP_F64 : NDZIP_HIP_F32;
        else if (k == "--extent") {
            for (const char *p = v.c_str(); *p && dims < 3;) {
                extent[dims++] = static_cast<uint32_t>(strtoul(p, const_cast<char **>(&p), 10));
                if (*p == ',') ++p;
            }
        }
    }
    if (dims == 0 || world == 0 || in_file.empty() || out_file.empty()) {
        fprintf(stderr, "usage: sharded_threads --world N --dtype f32|f64 --extent a[,b[,c]] --in ARRAY.bin --out STREAM.bin\n");
        return 2;
    }
    size_t bytes = dtype == NDZIP_HIP_F32 ? 4 : 8;
    for (int d = 0; d < dims; ++d) bytes *= extent[d];
    std::vector<char> array(bytes), stream;
    FILE *f = fopen(in_file.c_str(), "rb");
    if (!f || fread(array.data(), 1, bytes, f) != bytes) {
        fprintf(stderr, "cannot read %zu bytes from %s\n", bytes, in_file.c_str());
        return 2;
    }
    fclose(f);
    rendezvous rv;
    rv.world = world;
    rv.send.resize(world);
    std::vector<std::thread> ranks;
    for (uint32_t r = 0; r < world; ++r) ranks.emplace_back(run_rank, r, world, dtype, dims, extent, &array, &stream, &rv);
    for (auto &t : ranks) t.join();
    if (failures) return 1;
    f = fopen(out_file.c_str(), "wb");
    if (!f || fwrite(stream.data(), 1, stream.size(), f) != stream.size() || fclose(f) != 0) return 2;
    printf("%u ranks: stream of %zu bytes written\n", world, stream.size());
    return 0;
}
