// include/ndzip_hip_sharded.hh (the C++ adaptor of the sharded path) end to end on ONE shard: compress a raw array, compare the stream
// with a reference stream file byte for byte, decode the resident stream, load the reference stream into a second codec and decode it,
// and the adaptor's error behaviour (exceptions with the C ABI's message).  The exchange table is never called for one shard.
// Built against the kernels' functional model in the CPU suite and against the real library on the GPU box.
//   sharded_adaptor_roundtrip f32|f64 a[,b[,c]] ARRAY.bin STREAM.ref
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "ndzip_hip_sharded.hh"

namespace {

int never(void *, const uint32_t *, uint32_t *, size_t, void *) { return 1; }

std::vector<char> slurp(const char *path) {
    std::vector<char> v;
    FILE *f = fopen(path, "rb");
    if (!f) throw std::runtime_error(std::string("cannot open ") + path);
    char buf[1 << 16];
    for (size_t n; (n = fread(buf, 1, sizeof buf, f)) > 0;) v.insert(v.end(), buf, buf + n);
    fclose(f);
    return v;
}

template<typename T>
int run(const ndzip::extent &size, const char *array_path, const char *ref_path) {
    using W = ndzip::compressed_type<T>;
    const auto array = slurp(array_path), ref = slurp(ref_path);
    const ndzip_hip_collectives table{nullptr, never, nullptr};
    ndzip::hip_sharded_codec<T> codec(size, 0, 1, table), reader(size, 0, 1, table);
    if (codec.first_row() != 0 || codec.local_size()[0] != size[0] || codec.local_size().dimensions() != size.dimensions()) return 10;
    void *d_in = nullptr, *d_out = nullptr;
    if (hipMalloc(&d_in, array.size() ? array.size() : 1) != hipSuccess || hipMalloc(&d_out, array.size() ? array.size() : 1) != hipSuccess) return 11;
    if (hipMemcpy(d_in, array.data(), array.size(), hipMemcpyHostToDevice) != hipSuccess) return 12;

    bool threw = false;
    try {
        codec.decompress(static_cast<T *>(d_out));  // nothing compressed or loaded yet
    } catch (const std::runtime_error &e) {
        threw = strstr(e.what(), "nothing to decode") != nullptr;
    }
    if (!threw) return 13;

    codec.compress(static_cast<const T *>(d_in));
    codec.check();
    const auto lay = codec.stream_layout();
    if (lay.stream_words * sizeof(W) != ref.size()) return 14;
    std::vector<W> stream(lay.stream_words);
    codec.write_stream(stream.data(), stream.size(), true);
    if (memcmp(stream.data(), ref.data(), ref.size()) != 0) return 15;
    ndzip::index_type entries = 0;
    if (codec.header_global(&entries) == nullptr && entries != 0) return 16;

    std::vector<char> back(array.size());
    codec.decompress(static_cast<T *>(d_out));
    codec.check();
    if (hipMemcpy(back.data(), d_out, back.size(), hipMemcpyDeviceToHost) != hipSuccess || memcmp(back.data(), array.data(), array.size()) != 0) return 17;

    threw = false;
    try {
        reader.load(reinterpret_cast<const W *>(ref.data()), lay.stream_words > 3 ? lay.stream_words - 3 : 0);  // truncated
    } catch (const std::runtime_error &) {
        threw = true;
    }
    if (!threw && lay.stream_words > 3) return 18;
    reader.load(reinterpret_cast<const W *>(ref.data()), lay.stream_words);
    (void) hipMemsetAsync(d_out, 0xff, back.size() ? back.size() : 1, nullptr);
    reader.decompress(static_cast<T *>(d_out));
    reader.check();
    if (hipMemcpy(back.data(), d_out, back.size(), hipMemcpyDeviceToHost) != hipSuccess || memcmp(back.data(), array.data(), array.size()) != 0) return 19;

    threw = false;
    try {
        ndzip::hip_sharded_codec_nd<T, 1> wrong(ndzip::extent{64, 64}, 0, 1, table);  // <T, Dims> spelling: dimensionality is checked
    } catch (const std::runtime_error &e) {
        threw = strstr(e.what(), "dimensionality") != nullptr;
    }
    if (!threw) return 20;
    (void) hipFree(d_in);
    (void) hipFree(d_out);
    printf("sharded adaptor round trip ok: %llu words\n", static_cast<unsigned long long>(lay.stream_words));
    return 0;
}

}  // namespace

int main(int argc, char **argv) {
    if (argc != 5) return 2;
    ndzip::extent size(0);
    {
        ndzip::index_type c[3] = {0, 0, 0};
        int dims = 0;
        for (const char *p = argv[2]; *p && dims < 3;) {
            c[dims++] = static_cast<ndzip::index_type>(strtoul(p, const_cast<char **>(&p), 10));
            if (*p == ',') ++p;
        }
        size = ndzip::extent(dims);
        for (int d = 0; d < dims; ++d) size[d] = c[d];
    }
    try {
        return std::string(argv[1]) == "f64" ? run<double>(size, argv[3], argv[4]) : run<float>(size, argv[3], argv[4]);
    } catch (const std::exception &e) {
        fprintf(stderr, "unexpected exception: %s\n", e.what());
        return 1;
    }
}
