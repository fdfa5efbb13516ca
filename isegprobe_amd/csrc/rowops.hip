// Row-wise HBM-bound kernels on the token / pixel axis: LayerNorm (fp32 stats), the
// bilinear(align_corners=True) NHWC resize, the C->1 classifier, layout converters.
#include "isp_common.h"
#include <type_traits>

// ---------------------------------------------------------------------------------------
// LayerNorm over the last dim.  One wave per row; the row lives in registers (two-pass
// mean / variance exactly like torch: var = mean((x-mean)^2)), output bf16 or fp32.
// Row remap: out row r reads input row  r + (r / group_out) * skip + skip_first  where
// skip_first rows are dropped at the start of each group of group_in = group_out + skip
// rows (used to drop the cls token: DINOv2.py:533-534).
struct f16_t {  // IEEE-half storage tag (bf16_t is a plain unsigned short): same size, different conversion
    unsigned short bits;
};
template <typename T>
__device__ __forceinline__ float4 load4_as_float(const T* p) {
    if constexpr (sizeof(T) == 4) {
        return *reinterpret_cast<const float4*>(p);
    } else {
        const uint2 u = *reinterpret_cast<const uint2*>(p);
        if constexpr (std::is_same_v<T, f16_t>) return make_float4(h_lo(u.x), h_hi(u.x), h_lo(u.y), h_hi(u.y));
        else return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16),
                                __uint_as_float(u.y & 0xffff0000u));
    }
}
template <typename T>
__device__ __forceinline__ void store4_from_float(T* p, float4 o) {
    if constexpr (sizeof(T) == 4) *reinterpret_cast<float4*>(p) = o;
    else if constexpr (std::is_same_v<T, f16_t>) *reinterpret_cast<uint2*>(p) = make_uint2(pack2h_sat(o.x, o.y), pack2h_sat(o.z, o.w));
    else *reinterpret_cast<uint2*>(p) = make_uint2(pack2bf(o.x, o.y), pack2bf(o.z, o.w));
}

template <typename TIN, typename TOUT, int MAXV>  // MAXV: float4 chunks per lane
__global__ __launch_bounds__(256) void layernorm_kernel(const TIN* __restrict__ x, TOUT* __restrict__ y,
                                                         const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, long rows, int D, float eps,
                                                         int group_out, int skip, long ld_in, long ld_out) {
    const int lane = threadIdx.x & 63;
    const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const long rin = group_out > 0 ? r + (r / group_out + 1) * skip : r;
    const TIN* xr = x + rin * ld_in;
    const int nchunk = D >> 2;
    float4 v[MAXV];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = lane + i * 64;
        if (c < nchunk) {
            v[i] = load4_as_float(xr + c * 4);
            sum += (v[i].x + v[i].y) + (v[i].z + v[i].w);
        } else {
            v[i] = make_float4(0, 0, 0, 0);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    const float mean = sum / (float)D;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = lane + i * 64;
        if (c < nchunk) {
            const float a = v[i].x - mean, b = v[i].y - mean, cc = v[i].z - mean, d = v[i].w - mean;
            sq += (a * a + b * b) + (cc * cc + d * d);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
    const float rstd = 1.0f / sqrtf(sq / (float)D + eps);
    TOUT* yr = y + r * ld_out;
    for (int c = nchunk + lane; c < (int)(ld_out >> 2); c += 64) {  // zero the padding columns [D, ld_out)
        if constexpr (sizeof(TOUT) == 4) *reinterpret_cast<float4*>(yr + c * 4) = make_float4(0, 0, 0, 0);
        else *reinterpret_cast<uint2*>(yr + c * 4) = make_uint2(0, 0);
    }
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = lane + i * 64;
        if (c < nchunk) {
            const float4 g = *reinterpret_cast<const float4*>(gamma + c * 4);
            const float4 b = *reinterpret_cast<const float4*>(beta + c * 4);
            float4 o;
            o.x = (v[i].x - mean) * rstd * g.x + b.x;
            o.y = (v[i].y - mean) * rstd * g.y + b.y;
            o.z = (v[i].z - mean) * rstd * g.z + b.z;
            o.w = (v[i].w - mean) * rstd * g.w + b.w;
            store4_from_float(yr + c * 4, o);
        }
    }
}

template <typename TIN, typename TOUT>
static int launch_ln(const void* x, void* y, const float* g, const float* b, long rows, int D, float eps, int group_out,
                     int skip, long ld_in, long ld_out, hipStream_t s) {
    const int nchunk = D / 4;
    dim3 grid((unsigned)((rows + 3) / 4));
#define LN_CASE(MV)                                                                                                  \
    layernorm_kernel<TIN, TOUT, MV><<<grid, 256, 0, s>>>((const TIN*)x, (TOUT*)y, g, b, rows, D, eps, group_out, skip, \
                                                         ld_in, ld_out)
    if (nchunk <= 64) LN_CASE(1);
    else if (nchunk <= 128) LN_CASE(2);
    else if (nchunk <= 256) LN_CASE(4);
    else if (nchunk <= 512) LN_CASE(8);
    else return ISP_ERR_UNSUPPORTED;
#undef LN_CASE
    return isp_launch_status();
}

extern "C" int isp_layernorm_fwd(const void* x, void* y, const float* gamma, const float* beta, long rows, int D,
                                 float eps, int in_dtype, int out_dtype, int group_out, int skip, long ld_in,
                                 long ld_out, void* stream) {
    ISP_CHECK_ARG(x && y && gamma && beta && rows > 0 && D > 0 && D % 4 == 0 && group_out >= 0 && skip >= 0);
    if (ld_in <= 0) ld_in = D;
    if (ld_out <= 0) ld_out = D;
    ISP_CHECK_ARG(ld_in >= D && ld_out >= D && ld_in % 4 == 0 && ld_out % 4 == 0);
    hipStream_t s = (hipStream_t)stream;
    if (in_dtype == ISP_F32 && out_dtype == ISP_BF16) return launch_ln<float, bf16_t>(x, y, gamma, beta, rows, D, eps, group_out, skip, ld_in, ld_out, s);
    if (in_dtype == ISP_F32 && out_dtype == ISP_F32) return launch_ln<float, float>(x, y, gamma, beta, rows, D, eps, group_out, skip, ld_in, ld_out, s);
    if (in_dtype == ISP_BF16 && out_dtype == ISP_BF16) return launch_ln<bf16_t, bf16_t>(x, y, gamma, beta, rows, D, eps, group_out, skip, ld_in, ld_out, s);
    if (in_dtype == ISP_BF16 && out_dtype == ISP_F32) return launch_ln<bf16_t, float>(x, y, gamma, beta, rows, D, eps, group_out, skip, ld_in, ld_out, s);
    // IEEE-half rows (LoftUp's inference stream): from bf16 tokens, fp32 or half maps; to half
    if (in_dtype == ISP_BF16 && out_dtype == ISP_F16) return launch_ln<bf16_t, f16_t>(x, y, gamma, beta, rows, D, eps, group_out, skip, ld_in, ld_out, s);
    if (in_dtype == ISP_F16 && out_dtype == ISP_F16) return launch_ln<f16_t, f16_t>(x, y, gamma, beta, rows, D, eps, group_out, skip, ld_in, ld_out, s);
    if (in_dtype == ISP_F32 && out_dtype == ISP_F16) return launch_ln<float, f16_t>(x, y, gamma, beta, rows, D, eps, group_out, skip, ld_in, ld_out, s);
    return ISP_ERR_UNSUPPORTED;
}

// ---------------------------------------------------------------------------------------
// Bilinear resize, align_corners=True, NHWC bf16 -> NHWC bf16 (F.interpolate semantics of
// basic_upsamplers.py:28-33 / iseg_probe_model.py:120-129).  A thread owns 8 channels of
// one output pixel: 4 x 16-B loads, one 16-B store.  src = dst * (in-1)/(out-1) in fp32.
__global__ __launch_bounds__(256) void bilinear_nhwc_kernel(const bf16_t* __restrict__ in, bf16_t* __restrict__ out,
                                                             int h, int w, int H, int W, int C, float sy, float sx,
                                                             long total) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int cv = C >> 3;
    const int c8 = (int)(idx % cv);
    long pix = idx / cv;
    const int X = (int)(pix % W);
    pix /= W;
    const int Y = (int)(pix % H);
    const int b = (int)(pix / H);
    const float fy = sy * (float)Y, fx = sx * (float)X;
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = min(y0 + 1, h - 1), x1 = min(x0 + 1, w - 1);
    const float ly = fy - (float)y0, lx = fx - (float)x0;
    const float w00 = (1.f - ly) * (1.f - lx), w01 = (1.f - ly) * lx, w10 = ly * (1.f - lx), w11 = ly * lx;
    const bf16_t* base = in + (size_t)b * h * w * C + c8 * 8;
    const uint4 a = *reinterpret_cast<const uint4*>(base + ((size_t)y0 * w + x0) * C);
    const uint4 bq = *reinterpret_cast<const uint4*>(base + ((size_t)y0 * w + x1) * C);
    const uint4 c = *reinterpret_cast<const uint4*>(base + ((size_t)y1 * w + x0) * C);
    const uint4 d = *reinterpret_cast<const uint4*>(base + ((size_t)y1 * w + x1) * C);
    const unsigned* ua = &a.x;
    const unsigned* ub = &bq.x;
    const unsigned* uc = &c.x;
    const unsigned* ud = &d.x;
    unsigned r[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float lo = w00 * __uint_as_float(ua[i] << 16) + w01 * __uint_as_float(ub[i] << 16) +
                         w10 * __uint_as_float(uc[i] << 16) + w11 * __uint_as_float(ud[i] << 16);
        const float hi = w00 * __uint_as_float(ua[i] & 0xffff0000u) + w01 * __uint_as_float(ub[i] & 0xffff0000u) +
                         w10 * __uint_as_float(uc[i] & 0xffff0000u) + w11 * __uint_as_float(ud[i] & 0xffff0000u);
        r[i] = pack2bf(lo, hi);
    }
    *reinterpret_cast<uint4*>(out + idx * 8) = make_uint4(r[0], r[1], r[2], r[3]);
}

extern "C" int isp_resize_bilinear_ac_nhwc_bf16(const void* in, void* out, int B, int h, int w, int H, int W, int C,
                                                void* stream) {
    ISP_CHECK_ARG(in && out && B > 0 && h > 0 && w > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0);
    const float sy = H > 1 ? (float)(h - 1) / (float)(H - 1) : 0.f;
    const float sx = W > 1 ? (float)(w - 1) / (float)(W - 1) : 0.f;
    const long total = (long)B * H * W * (C / 8);
    bilinear_nhwc_kernel<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(
        (const bf16_t*)in, (bf16_t*)out, h, w, H, W, C, sy, sx, total);
    return isp_launch_status();
}

// Same resize on fp32 NCHW single/multi-channel maps (logits, probabilities, ROI crops):
// iseg_base_model.py:75-80, base_predictor.py:95-97, zoom_in.py:113-118,240-247.
__global__ __launch_bounds__(256) void bilinear_nchw_f32_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                                 int h, int w, int H, int W, float sy, float sx,
                                                                 long in_plane_stride, long total) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int X = (int)(idx % W);
    const long t = idx / W;
    const int Y = (int)(t % H);
    const long plane = t / H;
    const float fy = sy * (float)Y, fx = sx * (float)X;
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = min(y0 + 1, h - 1), x1 = min(x0 + 1, w - 1);
    const float ly = fy - (float)y0, lx = fx - (float)x0;
    const float* p = in + plane * in_plane_stride;
    // torch's upsample_bilinear2d accumulation order: h0*(w0*a + w1*b) + h1*(w0*c + w1*d)
    const float top = (1.f - lx) * p[(size_t)y0 * w + x0] + lx * p[(size_t)y0 * w + x1];
    const float bot = (1.f - lx) * p[(size_t)y1 * w + x0] + lx * p[(size_t)y1 * w + x1];
    out[idx] = (1.f - ly) * top + ly * bot;
}

extern "C" int isp_resize_bilinear_ac_nchw_f32(const float* in, float* out, long planes, int h, int w, int H, int W,
                                               long in_plane_stride, void* stream) {
    ISP_CHECK_ARG(in && out && planes > 0 && h > 0 && w > 0 && H > 0 && W > 0 && in_plane_stride >= (long)h * w);
    const float sy = H > 1 ? (float)(h - 1) / (float)(H - 1) : 0.f;
    const float sx = W > 1 ? (float)(w - 1) / (float)(W - 1) : 0.f;
    const long total = planes * H * W;
    bilinear_nchw_f32_kernel<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(
        in, out, h, w, H, W, sy, sx, in_plane_stride, total);
    return isp_launch_status();
}

// ---------------------------------------------------------------------------------------
// 1x1 classifier C -> 1 (BaseClassifierHead.classifier, heads/base_head.py:15): one dot
// product per pixel over an NHWC bf16 map, fp32 out.  16 lanes per pixel, 16-B loads.
__global__ __launch_bounds__(256) void classifier_kernel(const bf16_t* __restrict__ x, const float* __restrict__ wt,
                                                          float bias, float* __restrict__ out, long M, int C) {
    const int sub = threadIdx.x & 15;
    const long pix = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    float acc = 0.f;
    if (pix < M) {
        const bf16_t* xr = x + pix * C;
        for (int c = sub * 8; c < C; c += 128) {
            const uint4 u = *reinterpret_cast<const uint4*>(xr + c);
            const float4 w0 = *reinterpret_cast<const float4*>(wt + c);
            const float4 w1 = *reinterpret_cast<const float4*>(wt + c + 4);
            acc += __uint_as_float(u.x << 16) * w0.x + __uint_as_float(u.x & 0xffff0000u) * w0.y +
                   __uint_as_float(u.y << 16) * w0.z + __uint_as_float(u.y & 0xffff0000u) * w0.w +
                   __uint_as_float(u.z << 16) * w1.x + __uint_as_float(u.z & 0xffff0000u) * w1.y +
                   __uint_as_float(u.w << 16) * w1.z + __uint_as_float(u.w & 0xffff0000u) * w1.w;
        }
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if (pix < M && sub == 0) out[pix] = acc + bias;
}

extern "C" int isp_classifier_fwd(const void* x_nhwc_bf16, const float* weight, float bias, float* out, long M, int C,
                                  void* stream) {
    ISP_CHECK_ARG(x_nhwc_bf16 && weight && out && M > 0 && C > 0 && C % 8 == 0);
    const long threads = M * 16;
    classifier_kernel<<<(unsigned)((threads + 255) / 256), 256, 0, (hipStream_t)stream>>>(
        (const bf16_t*)x_nhwc_bf16, weight, bias, out, M, C);
    return isp_launch_status();
}

// ---------------------------------------------------------------------------------------
// Layout converters between the plugin API's NCHW fp32 tensors and the kernels' NHWC bf16.
__global__ __launch_bounds__(256) void nhwc_bf16_to_nchw_f32_kernel(const bf16_t* __restrict__ in,
                                                                     float* __restrict__ out, int C, long HW) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z;
    const long p0 = (long)blockIdx.x * 32;
    const int c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    for (int i = ty; i < 32; i += 8) {
        const long p = p0 + i;
        const int c = c0 + tx;
        tile[i][tx] = (p < HW && c < C) ? bf2f(in[((size_t)b * HW + p) * C + c]) : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int c = c0 + i;
        const long p = p0 + tx;
        if (p < HW && c < C) out[((size_t)b * C + c) * HW + p] = tile[tx][i];
    }
}

extern "C" int isp_nhwc_bf16_to_nchw_f32(const void* in, float* out, int B, int C, long HW, void* stream) {
    ISP_CHECK_ARG(in && out && B > 0 && C > 0 && HW > 0 && B <= 65535);
    dim3 grid((unsigned)((HW + 31) / 32), (C + 31) / 32, B);
    nhwc_bf16_to_nchw_f32_kernel<<<grid, 256, 0, (hipStream_t)stream>>>((const bf16_t*)in, out, C, HW);
    return isp_launch_status();
}

__global__ __launch_bounds__(256) void nchw_f32_to_nhwc_bf16_kernel(const float* __restrict__ in,
                                                                     bf16_t* __restrict__ out, int C, long HW,
                                                                     long sb, long sc, long sp) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z;
    const long p0 = (long)blockIdx.x * 32;
    const int c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int i = ty; i < 32; i += 8) {
        const int c = c0 + i;
        const long p = p0 + tx;
        tile[i][tx] = (p < HW && c < C) ? in[(size_t)b * sb + (size_t)c * sc + (size_t)p * sp] : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const long p = p0 + i;
        const int c = c0 + tx;
        if (p < HW && c < C) out[((size_t)b * HW + p) * C + c] = f2bf(tile[tx][i]);
    }
}

// in is addressed as in[b*sb + c*sc + p*sp] (element strides) so permuted NCHW views
// (the featurizer's [B,D,h,w] view of [B,h*w,D], DINOv2.py:545) need no copy first.
extern "C" int isp_nchw_f32_to_nhwc_bf16(const float* in, void* out, int B, int C, long HW, long stride_b,
                                         long stride_c, long stride_p, void* stream) {
    ISP_CHECK_ARG(in && out && B > 0 && C > 0 && HW > 0 && B <= 65535);
    dim3 grid((unsigned)((HW + 31) / 32), (C + 31) / 32, B);
    nchw_f32_to_nhwc_bf16_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(in, (bf16_t*)out, C, HW, stride_b, stride_c,
                                                                        stride_p);
    return isp_launch_status();
}

// ---------------------------------------------------------------------------------------
// Token add with optional cls skip: x[b, (has_cls ? 1 : 0) + t, :] += add[b, t, :]
// (click injection, DINOv2.py:516 and :523).  x f32 or bf16, add f32 or bf16.
template <typename TX, typename TA>
__global__ __launch_bounds__(256) void token_add_kernel(TX* __restrict__ x, const TA* __restrict__ add, long total4,
                                                         int T, int D4, int has_cls) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total4) return;
    const long row = idx / D4;
    const int c4 = (int)(idx - row * D4);
    const long b = row / T;
    const long xrow = has_cls ? row + b + 1 : row;
    float a[4];
    if constexpr (sizeof(TA) == 4) {
        const float4 v = *reinterpret_cast<const float4*>(add + idx * 4);
        a[0] = v.x, a[1] = v.y, a[2] = v.z, a[3] = v.w;
    } else {
        const uint2 u = *reinterpret_cast<const uint2*>(add + idx * 4);
        a[0] = __uint_as_float(u.x << 16), a[1] = __uint_as_float(u.x & 0xffff0000u);
        a[2] = __uint_as_float(u.y << 16), a[3] = __uint_as_float(u.y & 0xffff0000u);
    }
    TX* px = x + (xrow * D4 + c4) * 4;
    if constexpr (sizeof(TX) == 4) {
        float4 v = *reinterpret_cast<float4*>(px);
        v.x += a[0], v.y += a[1], v.z += a[2], v.w += a[3];
        *reinterpret_cast<float4*>(px) = v;
    } else {
        const uint2 u = *reinterpret_cast<const uint2*>(px);
        *reinterpret_cast<uint2*>(px) =
            make_uint2(pack2bf(__uint_as_float(u.x << 16) + a[0], __uint_as_float(u.x & 0xffff0000u) + a[1]),
                       pack2bf(__uint_as_float(u.y << 16) + a[2], __uint_as_float(u.y & 0xffff0000u) + a[3]));
    }
}

extern "C" int isp_token_add_fwd(void* x, int x_dtype, const void* add, int add_dtype, long B, int T, int D,
                                 int x_has_cls, void* stream) {
    ISP_CHECK_ARG(x && add && B > 0 && T > 0 && D > 0 && D % 4 == 0);
    const long total4 = B * T * (D / 4);
    const unsigned grid = (unsigned)((total4 + 255) / 256);
    hipStream_t s = (hipStream_t)stream;
#define TA_CASE(TX, TA) token_add_kernel<TX, TA><<<grid, 256, 0, s>>>((TX*)x, (const TA*)add, total4, T, D / 4, x_has_cls)
    if (x_dtype == ISP_F32 && add_dtype == ISP_F32) TA_CASE(float, float);
    else if (x_dtype == ISP_F32 && add_dtype == ISP_BF16) TA_CASE(float, bf16_t);
    else if (x_dtype == ISP_BF16 && add_dtype == ISP_F32) TA_CASE(bf16_t, float);
    else if (x_dtype == ISP_BF16 && add_dtype == ISP_BF16) TA_CASE(bf16_t, bf16_t);
    else return ISP_ERR_UNSUPPORTED;
#undef TA_CASE
    return isp_launch_status();
}

// ---------------------------------------------------------------------------------------
// Nearest and bicubic (align_corners=False, A=-0.75) NHWC bf16 resizes: F.interpolate of
// basic_upsamplers.py:18-25,36-42 and FeatUp JBU's bicubic x2 (third-party, see oracle).
// PyTorch source-index rules: nearest src = floor(dst * in/out); bicubic
// src = (dst + 0.5) * in/out - 0.5 with border-clamped taps.
__device__ __forceinline__ void cubic_coeffs(float t, float* w) {
    const float A = -0.75f;
    const float x0 = t + 1.f, x1 = t, x2 = 1.f - t, x3 = 2.f - t;
    w[0] = ((A * x0 - 5.f * A) * x0 + 8.f * A) * x0 - 4.f * A;
    w[1] = ((A + 2.f) * x1 - (A + 3.f)) * x1 * x1 + 1.f;
    w[2] = ((A + 2.f) * x2 - (A + 3.f)) * x2 * x2 + 1.f;
    w[3] = ((A * x3 - 5.f * A) * x3 + 8.f * A) * x3 - 4.f * A;
}

template <int MODE>  // 0 nearest, 1 bicubic
__global__ __launch_bounds__(256) void resize_nhwc_kernel(const bf16_t* __restrict__ in, bf16_t* __restrict__ out,
                                                           int h, int w, int H, int W, int C, float sy, float sx,
                                                           long total) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int cv = C >> 3;
    const int c8 = (int)(idx % cv);
    long pix = idx / cv;
    const int X = (int)(pix % W);
    pix /= W;
    const int Y = (int)(pix % H);
    const int b = (int)(pix / H);
    const bf16_t* base = in + (size_t)b * h * w * C + c8 * 8;
    if (MODE == 0) {
        const int ys = min((int)floorf((float)Y * sy), h - 1), xs = min((int)floorf((float)X * sx), w - 1);
        *reinterpret_cast<uint4*>(out + idx * 8) = *reinterpret_cast<const uint4*>(base + ((size_t)ys * w + xs) * C);
        return;
    }
    const float fy = sy * ((float)Y + 0.5f) - 0.5f, fx = sx * ((float)X + 0.5f) - 0.5f;
    const int iy = (int)floorf(fy), ix = (int)floorf(fx);
    float wy[4], wx[4];
    cubic_coeffs(fy - (float)iy, wy);
    cubic_coeffs(fx - (float)ix, wx);
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int yy = min(max(iy - 1 + i, 0), h - 1);
        float rowacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int xx = min(max(ix - 1 + j, 0), w - 1);
            const uint4 u = *reinterpret_cast<const uint4*>(base + ((size_t)yy * w + xx) * C);
            const unsigned* p = &u.x;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                rowacc[2 * k] += wx[j] * __uint_as_float(p[k] << 16);
                rowacc[2 * k + 1] += wx[j] * __uint_as_float(p[k] & 0xffff0000u);
            }
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] += wy[i] * rowacc[k];
    }
    *reinterpret_cast<uint4*>(out + idx * 8) =
        make_uint4(pack2bf(acc[0], acc[1]), pack2bf(acc[2], acc[3]), pack2bf(acc[4], acc[5]), pack2bf(acc[6], acc[7]));
}

extern "C" int isp_resize_nhwc_bf16(const void* in, void* out, int B, int h, int w, int H, int W, int C, int mode,
                                    void* stream) {
    ISP_CHECK_ARG(in && out && B > 0 && h > 0 && w > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0);
    if (mode == ISP_RESIZE_BILINEAR_AC) return isp_resize_bilinear_ac_nhwc_bf16(in, out, B, h, w, H, W, C, stream);
    const float sy = (float)h / (float)H, sx = (float)w / (float)W;
    const long total = (long)B * H * W * (C / 8);
    const unsigned grid = (unsigned)((total + 255) / 256);
    hipStream_t s = (hipStream_t)stream;
    if (mode == ISP_RESIZE_NEAREST)
        resize_nhwc_kernel<0><<<grid, 256, 0, s>>>((const bf16_t*)in, (bf16_t*)out, h, w, H, W, C, sy, sx, total);
    else if (mode == ISP_RESIZE_BICUBIC)
        resize_nhwc_kernel<1><<<grid, 256, 0, s>>>((const bf16_t*)in, (bf16_t*)out, h, w, H, W, C, sy, sx, total);
    else
        return ISP_ERR_UNSUPPORTED;
    return isp_launch_status();
}

// ---------------------------------------------------------------------------------------
// Click-map fusion epilogue of the predictor (reference core/inference/transforms/flip.py:32-36
// then base_transform.py:39): logits [2n,1,H,W] (second half = predictions on the mirrored
// image) -> probs [n,1,H,W] = sigmoid(0.5 * (a + flip_w(b))).  with_flip=0: probs = sigmoid(a).
__global__ __launch_bounds__(256) void fuse_flip_sigmoid_kernel(const float* __restrict__ logits,
                                                                 float* __restrict__ probs, int W, long plane,
                                                                 long total, int with_flip) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    float v = logits[idx];
    if (with_flip) {
        const long n = idx / plane, p = idx - n * plane;
        const long row = p / W;
        const int x = (int)(p - row * W);
        const float m = logits[total + n * plane + row * W + (W - 1 - x)];
        v = __fmul_rn(0.5f, __fadd_rn(v, m));  // 0.5 * (prob_map + flipped), as the reference orders it
    }
    probs[idx] = 1.0f / (1.0f + expf(-v));
}

extern "C" int isp_fuse_flip_sigmoid(const float* logits, float* probs, long n, int H, int W, int with_flip,
                                     void* stream) {
    ISP_CHECK_ARG(logits && probs && n > 0 && H > 0 && W > 0);
    const long plane = (long)H * W, total = n * plane;
    fuse_flip_sigmoid_kernel<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(logits, probs, W, plane,
                                                                                               total, with_flip);
    return isp_launch_status();
}


// ---------------------------------------------------------------------------------------
// out[m] = bias + sum_s partial[s][m]: closes the conv + classifier fusion (ISP_EP_RELU_DOT_PARTIAL_F32).
__global__ __launch_bounds__(256) void sum_partials_kernel(const float* __restrict__ partial, float* __restrict__ out,
                                                            long M, int slots, float bias) {
    const long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i >= M) return;
    if ((M & 3) == 0) {  // slot rows are 16-byte aligned only when M is a multiple of 4
        float4 acc = make_float4(bias, bias, bias, bias);
        for (int s = 0; s < slots; ++s) {
            const float4 v = *reinterpret_cast<const float4*>(partial + (size_t)s * M + i);
            acc.x += v.x, acc.y += v.y, acc.z += v.z, acc.w += v.w;
        }
        *reinterpret_cast<float4*>(out + i) = acc;
    } else {
        for (long j = i; j < M && j < i + 4; ++j) {
            float a = bias;
            for (int s = 0; s < slots; ++s) a += partial[(size_t)s * M + j];
            out[j] = a;
        }
    }
}

extern "C" int isp_sum_partials_f32(const float* partial, float* out, long M, int slots, float bias, void* stream) {
    ISP_CHECK_ARG(partial && out && M > 0 && slots > 0);
    sum_partials_kernel<<<(unsigned)(((M + 3) / 4 + 255) / 256), 256, 0, (hipStream_t)stream>>>(partial, out, M, slots, bias);
    return isp_launch_status();
}

// ---------------------------------------------------------------------------------------
// fp32-accurate products on the bf16 MFMA engine ("three bf16 products"): x = hi + lo with hi = bf16(x),
// lo = bf16(x - hi) (|x - hi - lo| <= 2^-17 |x|), so  x . w  ~=  hi.whi + hi.wlo + lo.whi  (the dropped lo.wlo term is
// 2^-18 relative).  Laid out along K the three products are ONE GEMM / conv of 3x the depth with fp32 accumulation:
// activations [hi | hi | lo], weights [whi | wlo | whi].  This kernel writes either layout from an fp32 matrix, with an
// optional activation (the fp32 epilogues of the engine carry none) and scale on the way in.  K is zero-padded to Kpad.
// Used by core/model/precise.py for the "logits within 1e-3 of the fp32 reference" gate; not on the bf16 product path.
__global__ __launch_bounds__(256) void split_bf16x3_kernel(const float* __restrict__ x, long ld_in, bf16_t* __restrict__ out,
                                                            long rows, int K, int Kpad, int weights, int act, float scale) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= rows * Kpad) return;
    const long r = idx / Kpad;
    const int k = (int)(idx - r * Kpad);
    float v = k < K ? x[r * ld_in + k] : 0.f;
    if (act == 1) v = fmaxf(v, 0.f);
    if (act == 2) v = 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
    if (act == 3) v = v / (1.0f + expf(-1.702f * v));  // CLIP's QuickGELU
    v *= scale;
    const bf16_t hi = f2bf(v);
    const bf16_t lo = f2bf(v - bf2f(hi));
    bf16_t* o = out + r * (3L * Kpad) + k;
    o[0] = hi;
    o[Kpad] = weights ? lo : hi;
    o[2L * Kpad] = weights ? hi : lo;
}

extern "C" int isp_split_bf16x3(const float* x, long ld_in, void* out_bf16, long rows, int K, int Kpad, int weights_layout,
                                int act, float scale, void* stream) {
    ISP_CHECK_ARG(x && out_bf16 && rows > 0 && K > 0 && Kpad >= K && ld_in >= K && act >= 0 && act <= 3);
    const long total = rows * Kpad;
    split_bf16x3_kernel<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(
        x, ld_in, (bf16_t*)out_bf16, rows, K, Kpad, weights_layout, act, scale);
    return isp_launch_status();
}

// in-place softmax over the first `cols` entries of every row of an fp32 [rows, ld] matrix; entries [cols, ld) become 0
// (zero-padded key columns).  One wave per row.
__global__ __launch_bounds__(256) void softmax_rows_kernel(float* __restrict__ x, long rows, int cols, long ld) {
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    float* p = x + row * ld;
    float m = -INFINITY;
    for (int c = lane; c < cols; c += 64) m = fmaxf(m, p[c]);
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    float s = 0.f;
    for (int c = lane; c < cols; c += 64) {
        const float e = expf(p[c] - m);
        p[c] = e;
        s += e;
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const float inv = 1.f / s;
    for (int c = lane; c < (int)ld; c += 64) p[c] = c < cols ? p[c] * inv : 0.f;
}

extern "C" int isp_softmax_rows_f32(float* x, long rows, int cols, long ld, void* stream) {
    ISP_CHECK_ARG(x && rows > 0 && cols > 0 && ld >= cols);
    softmax_rows_kernel<<<(unsigned)((rows + 3) / 4), 256, 0, (hipStream_t)stream>>>(x, rows, cols, ld);
    return isp_launch_status();
}
