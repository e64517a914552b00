// tests/cpp/adaptor_roundtrip.cc -- the reference's "decode(encode(input)) reproduces the input" test
// (src/test/codec_profile_test.inl:37-96) written against include/ndzip_hip.hh, i.e. what a reference user's code looks
// like after switching back-ends.  Built with g++ and linked against libndzip_hip.so by tests/test_cpp_adaptor.py.
#include <cstdio>
#include <cstring>
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

int main() {
    int rc = 0;
    rc |= run<float>(1, 4096 * 4 - 1);
    rc |= run<float>(2, 64 * 4 - 1) << 4;
    rc |= run<float>(3, 16 * 4 - 1) << 8;
    rc |= run<double>(1, 4096 * 4 - 1) << 12;
    rc |= run<double>(2, 64 * 4 - 1) << 16;
    rc |= run<double>(3, 16 * 4 - 1) << 20;
    std::printf(rc ? "FAILED 0x%x\n" : "adaptor round trips ok\n", rc);
    return rc != 0;
}
