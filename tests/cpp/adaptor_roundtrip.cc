// tests/cpp/adaptor_roundtrip.cc -- the reference's "decode(encode(input)) reproduces the input" test
// (src/test/codec_profile_test.inl:37-96) written against include/ndzip_hip.hh, i.e. what a reference user's code looks
// like after switching back-ends.  Built with g++ and linked against libndzip_hip.so by tests/test_cpp_adaptor.py.
#include <cstdio>
#include <cstring>
#include <memory>
#include <random>
#include <vector>

#include "ndzip_hip.hh"

template<typename T>
int run(ndzip::dim_type dims, ndzip::index_type n) {
    const auto size = ndzip::extent::broadcast(dims, n);
    std::vector<T> input(ndzip::num_elements(size));
    std::minstd_rand gen;
    std::uniform_real_distribution<T> dist;
    for (auto &v : input) v = dist(gen);
    std::fill(input.begin(), input.begin() + 8 * sizeof(T), T{});  // regression input of the reference test

    auto offloader = ndzip::make_hip_offloader<T>(dims);
    std::vector<ndzip::compressed_type<T>> stream(ndzip::hip_compressed_length_bound<T>(size));
    ndzip::kernel_duration dur{};
    stream.resize(offloader->compress(input.data(), size, stream.data(), &dur));
    std::vector<T> output(input.size());
    const auto read = offloader->decompress(stream.data(), static_cast<ndzip::index_type>(stream.size()), output.data(), size);
    if (read != stream.size()) return 1;
    if (std::memcmp(input.data(), output.data(), input.size() * sizeof(T)) != 0) return 2;
    if (dur.count() == 0) return 3;
    bool threw = false;
    try {
        offloader->compress(input.data(), ndzip::extent::broadcast(dims == 3 ? 2 : dims + 1, 4), stream.data());
    } catch (const std::runtime_error &) { threw = true; }
    return threw ? 0 : 4;
}

// the plain plugin interface (ndzip.hh:227-253): what a caller of make_compressor<T>(dims) / make_decompressor<T>(dims) does
template<typename T, ndzip::dim_type Dims>
int run_plugin(ndzip::index_type n) {
    const auto size = ndzip::extent::broadcast(Dims, n);
    std::vector<T> input(ndzip::num_elements(size));
    std::minstd_rand gen(11);
    std::uniform_real_distribution<T> dist;
    for (auto &v : input) v = dist(gen);
    std::unique_ptr<ndzip::compressor<T>> comp = ndzip::make_hip_host_compressor<T>(Dims);
    std::unique_ptr<ndzip::decompressor<T>> decomp = ndzip::make_hip_host_decompressor<T>(Dims);
    std::vector<ndzip::compressed_type<T>> stream(ndzip::hip_compressed_length_bound<T>(size));
    const auto words = comp->compress(input.data(), size, stream.data());
    std::vector<T> output(input.size());
    if (decomp->decompress(stream.data(), output.data(), size) != words) return 1;
    if (std::memcmp(input.data(), output.data(), input.size() * sizeof(T)) != 0) return 2;
    // the <T, Dims> spelling produces the same stream
    ndzip::hip_host_compressor_nd<T, Dims> comp_nd;
    std::vector<ndzip::compressed_type<T>> stream2(stream.size());
    if (comp_nd.compress(input.data(), size, stream2.data()) != words) return 3;
    if (std::memcmp(stream.data(), stream2.data(), words * sizeof(stream[0])) != 0) return 3;
    ndzip::hip_host_decompressor_nd<T, Dims> decomp_nd;
    std::fill(output.begin(), output.end(), T{});
    if (decomp_nd.decompress(stream.data(), output.data(), size) != words) return 4;
    if (std::memcmp(input.data(), output.data(), input.size() * sizeof(T)) != 0) return 4;
    bool threw = false;
    try {
        comp->compress(input.data(), ndzip::extent::broadcast(Dims == 3 ? 2 : Dims + 1, 4), stream.data());
    } catch (const std::runtime_error &) { threw = true; }
    if (!threw) return 5;
    // a corrupt header is an exception, not a device fault
    reinterpret_cast<uint32_t *>(stream.data())[0] = 0xfffffff0u;
    threw = false;
    try {
        decomp->decompress(stream.data(), output.data(), size);
    } catch (const std::runtime_error &) { threw = true; }
    return threw ? 0 : 6;
}

// several arrays in flight through the persistent offloader, pinned buffers; every stream must equal the one-shot offloader's
template<typename T>
int run_pipelined(ndzip::dim_type dims, ndzip::index_type n) {
    const auto size = ndzip::extent::broadcast(dims, n);
    const size_t count = ndzip::num_elements(size), bound = ndzip::hip_compressed_length_bound<T>(size);
    constexpr int slots = 3, jobs = 7;
    auto pipe = ndzip::make_hip_pipelined_offloader<T>(size, slots);
    auto oneshot = ndzip::make_hip_offloader<T>(dims);
    std::vector<std::unique_ptr<ndzip::hip_host_buffer<T>>> in;
    std::vector<std::unique_ptr<ndzip::hip_host_buffer<ndzip::compressed_type<T>>>> out;
    for (int s = 0; s < slots; ++s) {
        in.emplace_back(new ndzip::hip_host_buffer<T>(count));
        out.emplace_back(new ndzip::hip_host_buffer<ndzip::compressed_type<T>>(bound));
    }
    std::minstd_rand gen(7);
    std::uniform_real_distribution<T> dist;
    std::vector<std::vector<T>> inputs(jobs, std::vector<T>(count));
    for (auto &v : inputs) for (auto &x : v) x = dist(gen);
    std::vector<std::vector<ndzip::compressed_type<T>>> streams;
    auto retire = [&](int j) {
        const auto words = pipe->wait(j % slots);
        streams.emplace_back(out[j % slots]->data(), out[j % slots]->data() + words);
    };
    for (int j = 0; j < jobs; ++j) {
        if (j >= slots) retire(j - slots);
        std::memcpy(in[j % slots]->data(), inputs[j].data(), count * sizeof(T));
        pipe->submit_compress(j % slots, in[j % slots]->data(), size, out[j % slots]->data());
    }
    for (int j = jobs - slots; j < jobs; ++j) retire(j);
    std::vector<ndzip::compressed_type<T>> ref(bound);
    for (int j = 0; j < jobs; ++j) {
        const auto words = oneshot->compress(inputs[j].data(), size, ref.data());
        if (words != streams[j].size() || std::memcmp(ref.data(), streams[j].data(), words * sizeof(ref[0])) != 0) return 1;
    }
    // ... and it is an offloader<T>
    ndzip::offloader<T> &as_offloader = *pipe;
    std::vector<T> back(count);
    if (as_offloader.decompress(streams[2].data(), static_cast<ndzip::index_type>(streams[2].size()), back.data(), size) != streams[2].size()) return 2;
    return std::memcmp(back.data(), inputs[2].data(), count * sizeof(T)) != 0 ? 3 : 0;
}

int main() {
    int rc = 0;
    rc |= run<float>(1, 4096 * 4 - 1);
    rc |= run<float>(2, 64 * 4 - 1) << 4;
    rc |= run<float>(3, 16 * 4 - 1) << 8;
    rc |= run<double>(1, 4096 * 4 - 1) << 12;
    rc |= run<double>(2, 64 * 4 - 1) << 16;
    rc |= run<double>(3, 16 * 4 - 1) << 20;
    rc |= run_pipelined<float>(3, 16 * 4 - 1) << 24;
    rc |= run_pipelined<double>(2, 64 * 3 + 5) << 26;
    int rc2 = 0;
    rc2 |= run_plugin<float, 1>(4096 * 2 + 7);
    rc2 |= run_plugin<float, 3>(16 * 3 + 1) << 4;
    rc2 |= run_plugin<double, 2>(64 * 2 + 3) << 8;
    if (rc2) std::printf("plugin interface FAILED 0x%x\n", rc2);
    rc |= rc2 << 28;
    std::printf(rc ? "FAILED 0x%x\n" : "adaptor round trips ok\n", rc);
    return rc != 0;
}
