// tools/ldsbench2.hip -- ground truth for LDS read instructions on gfx950: cycles per wave-instruction (SQ_LDS_IDX_ACTIVE)
// for ds_read_b128 / ds_read2_b64 / ds_read_b64 / ds_read_b32 at several lane strides.  Inline asm so the instruction is
// what it says.  Run under rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE; 1 wavefront per launch,
// 16 instructions each.
#include <hip/hip_runtime.h>
#include <cstdio>

template<int Op>
__global__ void __launch_bounds__(64) k(uint32_t *out, uint32_t stride_bytes) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const uint32_t lane = threadIdx.x;
    for (uint32_t i = lane; i < 16384; i += 64) reinterpret_cast<uint32_t *>(smem)[i] = i;
    __syncthreads();
    const uint32_t addr = static_cast<uint32_t>(reinterpret_cast<uintptr_t>((const __attribute__((address_space(3))) char *) smem)) + lane * stride_bytes;
    uint32_t acc = 0;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        uint32_t a = addr + (j & 1) * 16;
        if (Op == 0) {
            uint4 v;
            asm volatile("ds_read_b128 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
            acc += v.x + v.w;
        } else if (Op == 1) {
            uint4 v;
            asm volatile("ds_read2_b64 %0, %1 offset1:1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
            acc += v.x + v.w;
        } else if (Op == 2) {
            uint2 v;
            asm volatile("ds_read_b64 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
            acc += v.x + v.y;
        } else {
            uint32_t v;
            asm volatile("ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
            acc += v;
        }
    }
    out[lane] = acc;
}

template<int Op> void run(uint32_t *out, uint32_t stride) {
    hipLaunchKernelGGL(k<Op>, dim3(1), dim3(64), 65536, 0, out, stride);
    hipDeviceSynchronize();
}
int main() {
    uint32_t *out;
    hipMalloc(&out, 4096);
    const uint32_t strides[] = {16, 144, 136, 272, 264, 80, 528};
    for (uint32_t s : strides) {
        run<0>(out, s);
        run<1>(out, s);
        run<2>(out, s);
        run<3>(out, s);
    }
    printf("order per stride {16,144,136,272,264,80,528}: b128, read2_b64, b64, b32\n");
    return 0;
}
