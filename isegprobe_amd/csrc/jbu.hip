// FeatUp joint-bilateral-upsampling (JBU) stage kernels for gfx950.
//
// The arithmetic lives in a third-party package that is absent from the reference tree
// (mhamilton723/FeatUp, reached through reference core/model/upsamplers/JBUFeatUp.py:30-32);
// it is restated from the published algorithm (JBULearnedRange / JBUStack / AdaptiveConv),
// see oracle/upsamplers.py::_jbu_stage.  One x2 stage =
//   guidance (fp32 NCHW, 3 ch) --adaptive_avg_pool--> G [B,3,GH,GW]
//   proj   = conv1x1(gelu(conv1x1(G)))                  [B,GH,GW,32]      (isp_jbu_range_proj)
//   kernel = softmax_t(temp * <proj(nbr_t), proj>) * gauss_t, renormalised,
//            += 0.1 * fixup_mlp([kernel, G])            [B,GH,GW,49] f32  (isp_jbu_kernels)
//   hr     = bicubic_x2(source)                         (isp_resize_nhwc_bf16)
//   out    = sum_t kernel_t * hr(reflect(p + t))        [B,GH,GW,C] bf16  (isp_jbu_adaptive_conv)
// All of it is stencil / per-pixel work: LDS-tiled where a neighbourhood is shared,
// coalesced 16-byte channel vectors on the feature maps.
#include "isp_common.h"

namespace {

constexpr int R = 3, DIA = 7, TAPS = 49, KEY = 32;

__device__ __forceinline__ int reflect(int i, int n) {  // F.pad(mode="reflect")
    i = i < 0 ? -i : i;
    return i >= n ? 2 * (n - 1) - i : i;
}

// --------------------------------------------------------------------------------------
// F.adaptive_avg_pool2d on NCHW fp32 planes (window = [floor(i*in/out), ceil((i+1)*in/out)) ).
__global__ __launch_bounds__(256) void adaptive_avg_pool_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                                 int H, int W, int OH, int OW, long total) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int ox = (int)(idx % OW);
    const long t = idx / OW;
    const int oy = (int)(t % OH);
    const long plane = t / OH;
    const int y0 = (int)(((long)oy * H) / OH), y1 = (int)((((long)oy + 1) * H + OH - 1) / OH);
    const int x0 = (int)(((long)ox * W) / OW), x1 = (int)((((long)ox + 1) * W + OW - 1) / OW);
    const float* p = in + plane * (long)H * W;
    float s = 0.f;
    for (int y = y0; y < y1; ++y)
        for (int x = x0; x < x1; ++x) s += p[(size_t)y * W + x];
    out[idx] = s / (float)((y1 - y0) * (x1 - x0));
}

// --------------------------------------------------------------------------------------
// range_proj: 1x1 (3 -> 32), GELU, 1x1 (32 -> 32).  One thread per pixel; output NHWC f32.
__global__ __launch_bounds__(256) void jbu_range_proj_kernel(const float* __restrict__ G, float* __restrict__ proj,
                                                              const float* __restrict__ w0, const float* __restrict__ b0,
                                                              const float* __restrict__ w3, const float* __restrict__ b3,
                                                              long HW, long total) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const long b = idx / HW, p = idx - b * HW;
    const float g0 = G[(b * 3 + 0) * HW + p], g1 = G[(b * 3 + 1) * HW + p], g2 = G[(b * 3 + 2) * HW + p];
    float hid[KEY];
#pragma unroll
    for (int j = 0; j < KEY; ++j) hid[j] = gelu_erf(b0[j] + w0[j * 3 + 0] * g0 + w0[j * 3 + 1] * g1 + w0[j * 3 + 2] * g2);
    float4* o = reinterpret_cast<float4*>(proj + idx * KEY);
#pragma unroll
    for (int m4 = 0; m4 < KEY / 4; ++m4) {
        float r[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int m = m4 * 4 + q;
            float s = b3[m];
#pragma unroll
            for (int j = 0; j < KEY; ++j) s += w3[m * KEY + j] * hid[j];
            r[q] = s;
        }
        o[m4] = make_float4(r[0], r[1], r[2], r[3]);
    }
}

// --------------------------------------------------------------------------------------
// Per-pixel 7x7 kernels.  Block = 16x16 pixels; the 22x22 reflect-padded proj tile is staged
// in LDS (pixel stride padded to 36 floats against bank conflicts).
constexpr int TS = 16, HALO = TS + 2 * R, PSTRIDE = 36;

__global__ __launch_bounds__(256) void jbu_kernels_kernel(const float* __restrict__ proj, const float* __restrict__ G,
                                                           float* __restrict__ kout, const float* __restrict__ f0w,
                                                           const float* __restrict__ f0b, const float* __restrict__ f3wT,
                                                           const float* __restrict__ f3b, float temp, float inv2s2,
                                                           int GH, int GW) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* tile = reinterpret_cast<float*>(smem);
    const int b = blockIdx.z, ty0 = blockIdx.y * TS, tx0 = blockIdx.x * TS;
    const long HW = (long)GH * GW;
    // stage proj tile (+halo, reflect) : HALO*HALO pixels x 8 float4
    for (int i = threadIdx.x; i < HALO * HALO * (KEY / 4); i += 256) {
        const int c4 = i % (KEY / 4), pix = i / (KEY / 4);
        const int py = pix / HALO, px = pix % HALO;
        const int gy = reflect(min(ty0 + py - R, GH - 1 + R), GH), gx = reflect(min(tx0 + px - R, GW - 1 + R), GW);
        *reinterpret_cast<float4*>(tile + pix * PSTRIDE + c4 * 4) =
            *reinterpret_cast<const float4*>(proj + ((size_t)b * HW + (size_t)gy * GW + gx) * KEY + c4 * 4);
    }
    __syncthreads();
    const int lx = threadIdx.x & 15, ly = threadIdx.x >> 4;
    const int y = ty0 + ly, x = tx0 + lx;
    if (y >= GH || x >= GW) return;

    float4 ctr[KEY / 4];
    const float* cp = tile + ((ly + R) * HALO + lx + R) * PSTRIDE;
#pragma unroll
    for (int c = 0; c < KEY / 4; ++c) ctr[c] = *reinterpret_cast<const float4*>(cp + c * 4);

    float k[TAPS];
    float mx = -INFINITY;
#pragma unroll
    for (int t = 0; t < TAPS; ++t) {
        const int i = t / DIA, j = t % DIA;
        const float* np = tile + ((ly + i) * HALO + lx + j) * PSTRIDE;
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < KEY / 4; ++c) {
            const float4 v = *reinterpret_cast<const float4*>(np + c * 4);
            s += v.x * ctr[c].x + v.y * ctr[c].y + v.z * ctr[c].z + v.w * ctr[c].w;
        }
        k[t] = s * temp;
        mx = fmaxf(mx, k[t]);
    }
    float sum = 0.f;
#pragma unroll
    for (int t = 0; t < TAPS; ++t) {
        k[t] = __expf(k[t] - mx);
        sum += k[t];
    }
    // softmax * spatial gaussian, renormalise (clamp 1e-7)
    const float inv = 1.f / sum;
    float sum2 = 0.f;
#pragma unroll
    for (int t = 0; t < TAPS; ++t) {
        const float dy = -1.f + (float)(t / DIA) * (2.f / (DIA - 1)), dx = -1.f + (float)(t % DIA) * (2.f / (DIA - 1));
        k[t] = k[t] * inv * __expf(-(dx * dx + dy * dy) * inv2s2);
        sum2 += k[t];
    }
    const float inv2 = 1.f / fmaxf(sum2, 1e-7f);
#pragma unroll
    for (int t = 0; t < TAPS; ++t) k[t] *= inv2;
    // fixup MLP on [k(49), G(3)]: 52 -> 49 (GELU) -> 49, added with weight 0.1
    const long p = (long)y * GW + x;
    const float g0 = G[((size_t)b * 3 + 0) * HW + p], g1 = G[((size_t)b * 3 + 1) * HW + p],
                g2 = G[((size_t)b * 3 + 2) * HW + p];
    float fix[TAPS];
#pragma unroll
    for (int m = 0; m < TAPS; ++m) fix[m] = f3b[m];
    for (int j = 0; j < TAPS; ++j) {  // hidden unit j (uniform loop: weights come through scalar loads)
        const float* wr = f0w + j * (TAPS + 3);
        float h = f0b[j] + wr[TAPS] * g0 + wr[TAPS + 1] * g1 + wr[TAPS + 2] * g2;
#pragma unroll
        for (int t = 0; t < TAPS; ++t) h += wr[t] * k[t];
        h = gelu_erf(h);
        const float* wc = f3wT + j * TAPS;  // column j of the second layer, stored transposed
#pragma unroll
        for (int m = 0; m < TAPS; ++m) fix[m] += wc[m] * h;
    }
    float* o = kout + ((size_t)b * HW + p) * TAPS;
#pragma unroll
    for (int t = 0; t < TAPS; ++t) o[t] = k[t] + 0.1f * fix[t];
}

// --------------------------------------------------------------------------------------
// Adaptive 7x7 convolution with reflect padding.  A thread owns 4 horizontally adjacent
// output pixels x 8 channels: each row of the window costs 10 16-byte loads for 4 x 7 taps
// (17.5 loads per output instead of 49); consecutive lanes are consecutive channel chunks of
// the same pixels, so the per-pixel kernel weights are wave-broadcast loads.
__global__ __launch_bounds__(256) void jbu_adaptive_conv_kernel(const bf16_t* __restrict__ hr,
                                                                 const float* __restrict__ kern,
                                                                 bf16_t* __restrict__ out, int GH, int GW, int C,
                                                                 long total) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int cv = C >> 3, gx = (GW + 3) >> 2;
    const int c8 = (int)(idx % cv);
    long t = idx / cv;
    const int x0 = (int)(t % gx) * 4;
    t /= gx;
    const int y = (int)(t % GH);
    const int b = (int)(t / GH);
    const bf16_t* base = hr + (size_t)b * GH * GW * C + c8 * 8;
    const float* kbase = kern + ((size_t)b * GH * GW + (size_t)y * GW) * TAPS;
    float acc[4][8];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int c = 0; c < 8; ++c) acc[p][c] = 0.f;
#pragma unroll 1
    for (int i = 0; i < DIA; ++i) {
        const bf16_t* rowp = base + (size_t)reflect(y + i - R, GH) * GW * C;
        float v[10][8];
#pragma unroll
        for (int q = 0; q < 10; ++q) {
            const int xs = reflect(min(x0 + q - R, GW - 1 + R), GW);
            const uint4 u = *reinterpret_cast<const uint4*>(rowp + (size_t)xs * C);
            const unsigned* w = &u.x;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[q][2 * e] = __uint_as_float(w[e] << 16);
                v[q][2 * e + 1] = __uint_as_float(w[e] & 0xffff0000u);
            }
        }
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const float* kp = kbase + (size_t)min(x0 + p, GW - 1) * TAPS + i * DIA;
#pragma unroll
            for (int j = 0; j < DIA; ++j) {
                const float wgt = kp[j];
#pragma unroll
                for (int c = 0; c < 8; ++c) acc[p][c] += wgt * v[p + j][c];
            }
        }
    }
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        if (x0 + p >= GW) break;
        *reinterpret_cast<uint4*>(out + (((size_t)b * GH + y) * GW + x0 + p) * C + c8 * 8) =
            make_uint4(pack2bf(acc[p][0], acc[p][1]), pack2bf(acc[p][2], acc[p][3]), pack2bf(acc[p][4], acc[p][5]),
                       pack2bf(acc[p][6], acc[p][7]));
    }
}

}  // namespace

extern "C" int isp_adaptive_avg_pool_nchw_f32(const float* in, float* out, long planes, int H, int W, int OH, int OW,
                                              void* stream) {
    ISP_CHECK_ARG(in && out && planes > 0 && H > 0 && W > 0 && OH > 0 && OW > 0);
    const long total = planes * OH * OW;
    adaptive_avg_pool_kernel<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(in, out, H, W, OH, OW,
                                                                                               total);
    return isp_launch_status();
}

extern "C" int isp_jbu_range_proj(const float* guidance, float* proj, const float* w0, const float* b0,
                                  const float* w3, const float* b3, int B, int GH, int GW, void* stream) {
    ISP_CHECK_ARG(guidance && proj && w0 && b0 && w3 && b3 && B > 0 && GH > 0 && GW > 0);
    const long HW = (long)GH * GW, total = HW * B;
    jbu_range_proj_kernel<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(guidance, proj, w0, b0, w3,
                                                                                            b3, HW, total);
    return isp_launch_status();
}

extern "C" int isp_jbu_kernels(const float* proj, const float* guidance, float* kernels, const float* fix0_w,
                               const float* fix0_b, const float* fix3_wT, const float* fix3_b, float range_temp,
                               float sigma_spatial, int B, int GH, int GW, void* stream) {
    ISP_CHECK_ARG(proj && guidance && kernels && fix0_w && fix0_b && fix3_wT && fix3_b && B > 0 && GH >= 4 && GW >= 4);
    ISP_CHECK_ARG(B <= 65535 && sigma_spatial != 0.f);
    const float temp = fminf(fmaxf(expf(range_temp), 1e-4f), 1e4f);
    const float inv2s2 = 1.0f / (2.f * sigma_spatial * sigma_spatial);
    const int lds = HALO * HALO * PSTRIDE * 4;
    static bool attr_done = false;
    if (!attr_done) {
        if (hipFuncSetAttribute((const void*)jbu_kernels_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds) !=
            hipSuccess)
            return ISP_ERR_LAUNCH;
        attr_done = true;
    }
    dim3 grid((GW + TS - 1) / TS, (GH + TS - 1) / TS, B);
    jbu_kernels_kernel<<<grid, 256, lds, (hipStream_t)stream>>>(proj, guidance, kernels, fix0_w, fix0_b, fix3_wT,
                                                                fix3_b, temp, inv2s2, GH, GW);
    return isp_launch_status();
}

extern "C" int isp_jbu_adaptive_conv(const void* hr_nhwc_bf16, const float* kernels, void* out_nhwc_bf16, int B,
                                     int GH, int GW, int C, void* stream) {
    ISP_CHECK_ARG(hr_nhwc_bf16 && kernels && out_nhwc_bf16 && B > 0 && GH >= 4 && GW >= 4 && C > 0 && C % 8 == 0);
    const long total = (long)B * GH * ((GW + 3) / 4) * (C / 8);
    jbu_adaptive_conv_kernel<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(
        (const bf16_t*)hr_nhwc_bf16, kernels, (bf16_t*)out_nhwc_bf16, GH, GW, C, total);
    return isp_launch_status();
}
