// LiFT image-pyramid kernels (reference core/model/upsamplers/LiFT.py:70-91,106-112): the small
// strided 3x3 convs on the guidance image (3->32, 32->32, stride 2, eval-mode BatchNorm folded,
// ReLU) and F.adaptive_max_pool2d.  Channel counts are tiny (<= 32), so these are direct VALU
// convolutions; the heavy part of LiFT (ConvTranspose as a GEMM, DoubleConv 3x3s, 1x1 out) runs
// on the MFMA GEMM / implicit-conv engine.
#include "isp_common.h"

namespace {

// out[b, oy, ox, n] = relu(bias[n] + sum_{ky,kx,c} in(b, 2oy-1+ky, 2ox-1+kx, c) * w[n][ky][kx][c]),
// zero padding 1.  Input either NCHW fp32 (IN_NCHW_F32) or NHWC bf16; output NHWC bf16 with
// COUT = 32.  One thread per output pixel; weights broadcast through scalar loads.
template <int CIN, bool IN_NCHW_F32, bool RELU>
__global__ __launch_bounds__(256) void conv3x3_s2_small_kernel(const void* __restrict__ in,
                                                                const float* __restrict__ w /* [32][3][3][CIN] */,
                                                                const float* __restrict__ bias, bf16_t* __restrict__ out,
                                                                int H, int W, int OH, int OW, long total) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int ox = (int)(idx % OW);
    const long t = idx / OW;
    const int oy = (int)(t % OH);
    const long b = t / OH;
    float acc[32];
#pragma unroll
    for (int n = 0; n < 32; ++n) acc[n] = bias[n];
#pragma unroll 1
    for (int ky = 0; ky < 3; ++ky) {
        const int iy = 2 * oy - 1 + ky;
        if ((unsigned)iy >= (unsigned)H) continue;
#pragma unroll 1
        for (int kx = 0; kx < 3; ++kx) {
            const int ix = 2 * ox - 1 + kx;
            if ((unsigned)ix >= (unsigned)W) continue;
            float v[CIN];
            if constexpr (IN_NCHW_F32) {
                const float* p = (const float*)in + (size_t)b * CIN * H * W + (size_t)iy * W + ix;
#pragma unroll
                for (int c = 0; c < CIN; ++c) v[c] = p[(size_t)c * H * W];
            } else {
                const bf16_t* p = (const bf16_t*)in + (((size_t)b * H + iy) * W + ix) * CIN;
#pragma unroll
                for (int c8 = 0; c8 < CIN / 8; ++c8) {
                    const uint4 u = *reinterpret_cast<const uint4*>(p + c8 * 8);
                    const unsigned* q = &u.x;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v[c8 * 8 + 2 * e] = __uint_as_float(q[e] << 16);
                        v[c8 * 8 + 2 * e + 1] = __uint_as_float(q[e] & 0xffff0000u);
                    }
                }
            }
            const float* wk = w + (ky * 3 + kx) * CIN;
#pragma unroll
            for (int n = 0; n < 32; ++n) {
                const float* wn = wk + (size_t)n * 9 * CIN;
#pragma unroll
                for (int c = 0; c < CIN; ++c) acc[n] += wn[c] * v[c];
            }
        }
    }
    bf16_t* o = out + idx * 32;
    auto act = [](float v) { return RELU ? fmaxf(v, 0.f) : v; };  // RELU off: the raw conv, ahead of a train-mode BatchNorm
#pragma unroll
    for (int n8 = 0; n8 < 4; ++n8)
        *reinterpret_cast<uint4*>(o + n8 * 8) =
            make_uint4(pack2bf(act(acc[n8 * 8 + 0]), act(acc[n8 * 8 + 1])), pack2bf(act(acc[n8 * 8 + 2]), act(acc[n8 * 8 + 3])),
                       pack2bf(act(acc[n8 * 8 + 4]), act(acc[n8 * 8 + 5])), pack2bf(act(acc[n8 * 8 + 6]), act(acc[n8 * 8 + 7])));
}

// F.adaptive_max_pool2d on NHWC bf16 (window = [floor(i*in/out), ceil((i+1)*in/out)) ), 8 ch / thread
__global__ __launch_bounds__(256) void adaptive_max_pool_nhwc_kernel(const bf16_t* __restrict__ in,
                                                                      bf16_t* __restrict__ out, int H, int W, int OH,
                                                                      int OW, int C, long total) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int cv = C >> 3, c8 = (int)(idx % cv);
    long t = idx / cv;
    const int ox = (int)(t % OW);
    t /= OW;
    const int oy = (int)(t % OH);
    const long b = t / OH;
    const int y0 = (int)(((long)oy * H) / OH), y1 = (int)((((long)oy + 1) * H + OH - 1) / OH);
    const int x0 = (int)(((long)ox * W) / OW), x1 = (int)((((long)ox + 1) * W + OW - 1) / OW);
    float m[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) m[e] = -INFINITY;
    for (int y = y0; y < y1; ++y)
        for (int x = x0; x < x1; ++x) {
            const uint4 u = *reinterpret_cast<const uint4*>(in + (((size_t)b * H + y) * W + x) * C + c8 * 8);
            const unsigned* q = &u.x;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                m[2 * e] = fmaxf(m[2 * e], __uint_as_float(q[e] << 16));
                m[2 * e + 1] = fmaxf(m[2 * e + 1], __uint_as_float(q[e] & 0xffff0000u));
            }
        }
    *reinterpret_cast<uint4*>(out + idx * 8) =
        make_uint4(pack2bf(m[0], m[1]), pack2bf(m[2], m[3]), pack2bf(m[4], m[5]), pack2bf(m[6], m[7]));
}

}  // namespace

extern "C" int isp_conv3x3_s2_c32(const void* in, int in_is_nchw_f32, int cin, const float* w, const float* bias,
                                  void* out_nhwc_bf16, int B, int H, int W, int relu, void* stream) {
    ISP_CHECK_ARG(in && w && bias && out_nhwc_bf16 && B > 0 && H > 0 && W > 0);
    const int OH = (H + 1) / 2, OW = (W + 1) / 2;  // k3 s2 p1
    const long total = (long)B * OH * OW;
    const unsigned grid = (unsigned)((total + 255) / 256);
    hipStream_t s = (hipStream_t)stream;
    bf16_t* o = (bf16_t*)out_nhwc_bf16;
    if (in_is_nchw_f32 && cin == 3) {
        if (relu) conv3x3_s2_small_kernel<3, true, true><<<grid, 256, 0, s>>>(in, w, bias, o, H, W, OH, OW, total);
        else conv3x3_s2_small_kernel<3, true, false><<<grid, 256, 0, s>>>(in, w, bias, o, H, W, OH, OW, total);
    } else if (!in_is_nchw_f32 && cin == 32) {
        if (relu) conv3x3_s2_small_kernel<32, false, true><<<grid, 256, 0, s>>>(in, w, bias, o, H, W, OH, OW, total);
        else conv3x3_s2_small_kernel<32, false, false><<<grid, 256, 0, s>>>(in, w, bias, o, H, W, OH, OW, total);
    } else {
        return ISP_ERR_UNSUPPORTED;
    }
    return isp_launch_status();
}

extern "C" int isp_adaptive_max_pool_nhwc_bf16(const void* in, void* out, int B, int H, int W, int OH, int OW, int C,
                                               void* stream) {
    ISP_CHECK_ARG(in && out && B > 0 && H > 0 && W > 0 && OH > 0 && OW > 0 && C > 0 && C % 8 == 0);
    const long total = (long)B * OH * OW * (C / 8);
    adaptive_max_pool_nhwc_kernel<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(
        (const bf16_t*)in, (bf16_t*)out, H, W, OH, OW, C, total);
    return isp_launch_status();
}
