// On-box roofline probes ("confirm both peaks with an on-box microbenchmark"): a register-resident
// bf16 MFMA loop and a float4 copy.  Diagnostics only -- nothing on the product path calls them; tools/peaks.py prints
// the numbers DESIGN.md quotes next to the nominal 2.5 PFLOP/s / 8 TB/s.
#include "isp_common.h"

namespace {

// 4 waves per block, 16 independent accumulator tiles per wave, operands in registers (loaded once from `seed` so the
// data is whatever the caller put there: zeros show the unthrottled clock, random bits the clock under real toggling)
__global__ __launch_bounds__(256) void mfma_peak_kernel(const bf16_t* __restrict__ seed, float* __restrict__ sink, int iters) {
    const int lane = threadIdx.x & 63;
    bf16x8 a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        a[i] = *reinterpret_cast<const bf16x8*>(seed + ((size_t)(threadIdx.x * 8 + i) * 8) % 65536);
        b[i] = *reinterpret_cast<const bf16x8*>(seed + ((size_t)(threadIdx.x * 8 + 4 + i) * 8) % 65536);
    }
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][3];
    if (s == 12345.678f) sink[blockIdx.x * 256 + threadIdx.x] = s;  // keeps the loop alive, practically never taken
    if (lane == 0 && blockIdx.x == 0 && threadIdx.x == 0) sink[0] = s;
}

// the same loop on v_mfma_f32_32x32x16_bf16 (half the operand-register bytes per MAC): 4 independent 32x32 tiles
__global__ __launch_bounds__(256) void mfma_peak32_kernel(const bf16_t* __restrict__ seed, float* __restrict__ sink, int iters) {
    bf16x8 a[2], b[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        a[i] = *reinterpret_cast<const bf16x8*>(seed + ((size_t)(threadIdx.x * 8 + i) * 8) % 65536);
        b[i] = *reinterpret_cast<const bf16x8*>(seed + ((size_t)(threadIdx.x * 8 + 4 + i) * 8) % 65536);
    }
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 2; ++rep)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) s += acc[i][j][0] + acc[i][j][15];
    if (s == 12345.678f) sink[blockIdx.x * 256 + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) sink[1] = s;
}

__global__ __launch_bounds__(256) void copy_peak_kernel(const float4* __restrict__ src, float4* __restrict__ dst, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) dst[i] = src[i];
}

}  // namespace

extern "C" int isp_probe_mfma_bf16(const void* seed_bf16_64k, float* sink, int blocks, int iters, void* stream) {
    ISP_CHECK_ARG(seed_bf16_64k && sink && blocks > 0 && iters > 0);
    mfma_peak_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>((const bf16_t*)seed_bf16_64k, sink, iters);
    return isp_launch_status();
}

extern "C" int isp_probe_mfma_bf16_32x32(const void* seed_bf16_64k, float* sink, int blocks, int iters, void* stream) {
    ISP_CHECK_ARG(seed_bf16_64k && sink && blocks > 0 && iters > 0);
    mfma_peak32_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>((const bf16_t*)seed_bf16_64k, sink, iters);
    return isp_launch_status();
}

extern "C" int isp_probe_copy(const void* src, void* dst, long bytes, void* stream) {
    ISP_CHECK_ARG(src && dst && bytes > 0 && bytes % 16 == 0);
    copy_peak_kernel<<<256 * 16, 256, 0, (hipStream_t)stream>>>((const float4*)src, (float4*)dst, bytes / 16);
    return isp_launch_status();
}
