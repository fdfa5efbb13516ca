// FeatUp JBU in plain fp32, stage by stage as the published algorithm states it (JBULearnedRange.forward of
// mhamilton723/FeatUp, called from core/model/upsamplers/JBUFeatUp.py:30-32 of the reference): per-pixel 49-tap kernel
// (range softmax x spatial Gaussian + 0.1 * fix-up MLP), bicubic x2 of the source, reflect-padded 7x7 adaptive
// convolution.  NOT the product path -- that is the composite-kernel / MFMA formulation of jbu.hip -- but the fp32
// checking mode (core/model/precise.py) and, being a separate derivation, an on-device cross-check of the composite
// formulation.  Simple one-thread-per-output kernels, accurate libm (expf / erff), no LDS tiling.
#include "isp_common.h"

namespace {

constexpr int KEY = 32, R = 3, DIA = 7, TAPS = 49;

__device__ __forceinline__ int reflect_i(int i, int n) {  // F.pad(mode="reflect")
    i = i < 0 ? -i : i;
    return i >= n ? 2 * (n - 1) - i : i;
}

// k[b,y,x,49]: softmax_t(temp * <proj(p), proj(reflect(p+t))>) * gauss_t, renormalised (clamp 1e-7), plus
// 0.1 * W3 gelu(W0 [k; guidance(p)] + b0) + b3.   proj [B,GH,GW,32] f32, guidance [B,3,GH,GW] f32 (the pooled image).
__global__ __launch_bounds__(64) void jbu_kernels_f32_kernel(const float* __restrict__ proj, const float* __restrict__ G,
                                                             float* __restrict__ kout, const float* __restrict__ f0w,
                                                             const float* __restrict__ f0b, const float* __restrict__ f3w,
                                                             const float* __restrict__ f3b, float temp, float inv2s2,
                                                             int GH, int GW, long total) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const long HW = (long)GH * GW;
    const long b = idx / HW, p = idx - b * HW;
    const int y = (int)(p / GW), x = (int)(p - (long)y * GW);
    const float* pb = proj + b * HW * KEY;
    float ctr[KEY];
#pragma unroll
    for (int c = 0; c < KEY; ++c) ctr[c] = pb[p * KEY + c];
    float k[TAPS];
    float mx = -INFINITY;
    for (int t = 0; t < TAPS; ++t) {
        const int yy = reflect_i(y + t / DIA - R, GH), xx = reflect_i(x + t % DIA - R, GW);
        const float* q = pb + ((long)yy * GW + xx) * KEY;
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < KEY; ++c) s = fmaf(q[c], ctr[c], s);
        k[t] = s * temp;
        mx = fmaxf(mx, k[t]);
    }
    float sum = 0.f;
    for (int t = 0; t < TAPS; ++t) {
        k[t] = expf(k[t] - mx);
        sum += k[t];
    }
    float sum2 = 0.f;
    for (int t = 0; t < TAPS; ++t) {
        const float dy = -1.f + (float)(t / DIA) * (2.f / (DIA - 1)), dx = -1.f + (float)(t % DIA) * (2.f / (DIA - 1));
        k[t] = (k[t] / sum) * expf(-(dx * dx + dy * dy) * inv2s2);
        sum2 += k[t];
    }
    const float inv2 = 1.f / fmaxf(sum2, 1e-7f);
    for (int t = 0; t < TAPS; ++t) k[t] *= inv2;
    const float g[3] = {G[(b * 3 + 0) * HW + p], G[(b * 3 + 1) * HW + p], G[(b * 3 + 2) * HW + p]};
    float hid[TAPS];
    for (int j = 0; j < TAPS; ++j) {  // fixup_proj[0]: [49, 52] over [k(49), guidance(3)]
        const float* wr = f0w + j * (TAPS + 3);
        float s = f0b[j];
        for (int i = 0; i < TAPS; ++i) s = fmaf(wr[i], k[i], s);
        s = fmaf(wr[TAPS], g[0], fmaf(wr[TAPS + 1], g[1], fmaf(wr[TAPS + 2], g[2], s)));
        hid[j] = 0.5f * s * (1.0f + erff(s * 0.70710678118654752440f));
    }
    float* o = kout + idx * TAPS;
    for (int m = 0; m < TAPS; ++m) {  // fixup_proj[3]: [49, 49]
        const float* wr = f3w + m * TAPS;
        float s = f3b[m];
        for (int j = 0; j < TAPS; ++j) s = fmaf(wr[j], hid[j], s);
        o[m] = k[m] + 0.1f * s;
    }
}

// F.interpolate(mode="bicubic", align_corners=False) by exactly x2 on NHWC fp32 (A = -0.75, indices clamped)
__device__ __forceinline__ void cubic_w(float t, float (&w)[4]) {
    const float A = -0.75f;
    const float x0 = t + 1.f, x1 = t, x2 = 1.f - t, x3 = 2.f - t;
    w[0] = ((A * x0 - 5.f * A) * x0 + 8.f * A) * x0 - 4.f * A;
    w[1] = ((A + 2.f) * x1 - (A + 3.f)) * x1 * x1 + 1.f;
    w[2] = ((A + 2.f) * x2 - (A + 3.f)) * x2 * x2 + 1.f;
    w[3] = ((A * x3 - 5.f * A) * x3 + 8.f * A) * x3 - 4.f * A;
}

__global__ __launch_bounds__(256) void bicubic_x2_f32_kernel(const float* __restrict__ src, float* __restrict__ out, int h, int w,
                                                             int C, long total) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int cv = C >> 2;
    const int c4 = (int)(idx % cv);
    long pix = idx / cv;
    const int X = (int)(pix % (2 * w));
    pix /= 2 * w;
    const int Y = (int)(pix % (2 * h));
    const long b = pix / (2 * h);
    const float fy = ((float)Y + 0.5f) * 0.5f - 0.5f, fx = ((float)X + 0.5f) * 0.5f - 0.5f;
    const float flY = floorf(fy), flX = floorf(fx);
    float wy[4], wx[4];
    cubic_w(fy - flY, wy);
    cubic_w(fx - flX, wx);
    const int iy = (int)flY, ix = (int)flX;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int yy = min(max(iy - 1 + i, 0), h - 1);
        float4 row = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int xx = min(max(ix - 1 + j, 0), w - 1);
            const float4 v = *reinterpret_cast<const float4*>(src + ((b * h + yy) * (long)w + xx) * C + c4 * 4);
            row.x = fmaf(wx[j], v.x, row.x), row.y = fmaf(wx[j], v.y, row.y);
            row.z = fmaf(wx[j], v.z, row.z), row.w = fmaf(wx[j], v.w, row.w);
        }
        acc.x = fmaf(wy[i], row.x, acc.x), acc.y = fmaf(wy[i], row.y, acc.y);
        acc.z = fmaf(wy[i], row.z, acc.z), acc.w = fmaf(wy[i], row.w, acc.w);
    }
    *reinterpret_cast<float4*>(out + idx * 4) = acc;
}

// AdaptiveConv: out[b,y,x,c] = sum_{i,j} hr[b, reflect(y+i-3), reflect(x+j-3), c] * k[b,y,x,i*7+j]
__global__ __launch_bounds__(256) void adaptive_conv7_f32_kernel(const float* __restrict__ hr, const float* __restrict__ k,
                                                                 float* __restrict__ out, int GH, int GW, int C, long total) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int cv = C >> 2;
    const int c4 = (int)(idx % cv);
    long pix = idx / cv;
    const int X = (int)(pix % GW);
    const long t2 = pix / GW;
    const int Y = (int)(t2 % GH);
    const long b = t2 / GH;
    const float* kp = k + pix * TAPS;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int i = 0; i < DIA; ++i) {
        const int yy = reflect_i(Y + i - R, GH);
        for (int j = 0; j < DIA; ++j) {
            const int xx = reflect_i(X + j - R, GW);
            const float wgt = kp[i * DIA + j];
            const float4 v = *reinterpret_cast<const float4*>(hr + ((b * GH + yy) * (long)GW + xx) * C + c4 * 4);
            acc.x = fmaf(wgt, v.x, acc.x), acc.y = fmaf(wgt, v.y, acc.y);
            acc.z = fmaf(wgt, v.z, acc.z), acc.w = fmaf(wgt, v.w, acc.w);
        }
    }
    *reinterpret_cast<float4*>(out + idx * 4) = acc;
}

}  // namespace

extern "C" int isp_jbu_kernels_f32(const float* proj, const float* guidance, float* k_out, const float* fix0_w,
                                   const float* fix0_b, const float* fix3_w, const float* fix3_b, float range_temp,
                                   float sigma_spatial, int B, int GH, int GW, void* stream) {
    ISP_CHECK_ARG(proj && guidance && k_out && fix0_w && fix0_b && fix3_w && fix3_b && B > 0 && GH >= 4 && GW >= 4);
    ISP_CHECK_ARG(sigma_spatial != 0.f);
    const float temp = fminf(fmaxf(expf(range_temp), 1e-4f), 1e4f);
    const float inv2s2 = 1.0f / (2.f * sigma_spatial * sigma_spatial);
    const long total = (long)B * GH * GW;
    jbu_kernels_f32_kernel<<<(unsigned)((total + 63) / 64), 64, 0, (hipStream_t)stream>>>(
        proj, guidance, k_out, fix0_w, fix0_b, fix3_w, fix3_b, temp, inv2s2, GH, GW, total);
    return isp_launch_status();
}

extern "C" int isp_bicubic_x2_nhwc_f32(const float* src, float* out, int B, int h, int w, int C, void* stream) {
    ISP_CHECK_ARG(src && out && B > 0 && h > 0 && w > 0 && C > 0 && C % 4 == 0);
    const long total = (long)B * 4 * h * w * (C / 4);
    bicubic_x2_f32_kernel<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(src, out, h, w, C, total);
    return isp_launch_status();
}

extern "C" int isp_adaptive_conv7_nhwc_f32(const float* hr, const float* k49, float* out, int B, int GH, int GW, int C,
                                           void* stream) {
    ISP_CHECK_ARG(hr && k49 && out && B > 0 && GH >= 4 && GW >= 4 && C > 0 && C % 4 == 0);
    const long total = (long)B * GH * GW * (C / 4);
    adaptive_conv7_f32_kernel<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(hr, k49, out, GH, GW, C, total);
    return isp_launch_status();
}
