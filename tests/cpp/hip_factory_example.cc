// tests/cpp/hip_factory_example.cc -- the translation unit INTEGRATION.md section 2 proposes for the reference tree
// (src/ndzip/hip_factory.cc), compiled against the reference's own headers by tests/test_cpp_adaptor.py.
#include <cassert>
#include <stdexcept>
#define NDZIP_HIP_WITH_REFERENCE_HEADERS 1  // adaptor uses ndzip::extent / offloader<T> / compressor_requirements
#include <ndzip_hip.hh>
template std::unique_ptr<ndzip::offloader<float>> ndzip::make_hip_offloader<float>(ndzip::dim_type);
template std::unique_ptr<ndzip::offloader<double>> ndzip::make_hip_offloader<double>(ndzip::dim_type);
// ... and the plain plugin interface: what the reference's make_compressor<T> / make_decompressor<T> (cpu_factory.cc:25-49) would
// return for a HIP target
template std::unique_ptr<ndzip::compressor<float>> ndzip::make_hip_host_compressor<float>(ndzip::dim_type, unsigned);
template std::unique_ptr<ndzip::compressor<double>> ndzip::make_hip_host_compressor<double>(ndzip::dim_type, unsigned);
template std::unique_ptr<ndzip::decompressor<float>> ndzip::make_hip_host_decompressor<float>(ndzip::dim_type, unsigned);
template std::unique_ptr<ndzip::decompressor<double>> ndzip::make_hip_host_decompressor<double>(ndzip::dim_type, unsigned);
