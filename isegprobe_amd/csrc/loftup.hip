// LoftUp front end (reference core/model/upsamplers/loftup/layers.py:61-158, loftup.py:48-56):
// batch-global MinMaxScaler statistics and the fused Fourier-feature + ChannelNorm producer of
// the first 3x3 conv's input.  Everything downstream (convs, LayerNorms, projections, fused
// cross-attention, feed-forward) reuses the GEMM / conv / attention / LayerNorm kernels.
#include "isp_common.h"

namespace {

// ---- per-channel min / max over (batch, H, W) of an NCHW fp32 tensor, two stages.
__global__ __launch_bounds__(256) void minmax_partial_kernel(const float* __restrict__ x, float* __restrict__ part,
                                                              int C, long HW, int nblk) {
    const int c = blockIdx.y, b = blockIdx.z;
    const float* p = x + ((size_t)b * C + c) * HW;
    float lo = INFINITY, hi = -INFINITY;
    for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4; i < HW; i += (long)nblk * 1024) {
        if (i + 3 < HW) {
            const float4 v = *reinterpret_cast<const float4*>(p + i);
            lo = fminf(fminf(lo, fminf(v.x, v.y)), fminf(v.z, v.w));
            hi = fmaxf(fmaxf(hi, fmaxf(v.x, v.y)), fmaxf(v.z, v.w));
        } else {
            for (long j = i; j < HW; ++j) lo = fminf(lo, p[j]), hi = fmaxf(hi, p[j]);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) lo = fminf(lo, __shfl_xor(lo, o)), hi = fmaxf(hi, __shfl_xor(hi, o));
    __shared__ float slo[4], shi[4];
    if ((threadIdx.x & 63) == 0) slo[threadIdx.x >> 6] = lo, shi[threadIdx.x >> 6] = hi;
    __syncthreads();
    if (threadIdx.x == 0) {
        const size_t o = (((size_t)c * gridDim.z + b) * nblk + blockIdx.x) * 2;
        part[o] = fminf(fminf(slo[0], slo[1]), fminf(slo[2], slo[3]));
        part[o + 1] = fmaxf(fmaxf(shi[0], shi[1]), fmaxf(shi[2], shi[3]));
    }
}

__global__ __launch_bounds__(64) void minmax_final_kernel(const float* __restrict__ part, float* __restrict__ out,
                                                           int n) {
    const int c = blockIdx.x;
    float lo = INFINITY, hi = -INFINITY;
    for (int i = threadIdx.x; i < n; i += 64) {
        lo = fminf(lo, part[((size_t)c * n + i) * 2]);
        hi = fmaxf(hi, part[((size_t)c * n + i) * 2 + 1]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) lo = fminf(lo, __shfl_xor(lo, o)), hi = fmaxf(hi, __shfl_xor(hi, o));
    if (threadIdx.x == 0) out[c * 2] = lo, out[c * 2 + 1] = hi;
}

// torch.linspace(-1, 1, n)[i] in fp32 (symmetric evaluation, as ATen does it)
__device__ __forceinline__ float linspace_pm1(int i, int n) {
    if (n == 1) return -1.f;
    const float step = 2.f / (float)(n - 1);
    return i < n / 2 ? -1.f + step * (float)i : 1.f - step * (float)(n - 1 - i);
}

// ---- Fourier features + ChannelNorm.  One wave per pixel; lane l owns feature channels
// l, l+64, l+128, l+192 of the 2*5*F+3 (= 203 for F = 20); mean / variance by wave reduction;
// output row of `ldo` bf16 (channels past 203 zero-filled).
//   feats5 = [grid_h, grid_w, c0, c1, c2],  ci = (img - lo_i) / max(hi_i - lo_i, 1e-4) - 0.5
//   sin block: idx = f*5 + m -> sin(feats5[m] * freq[f] + bias_sin[idx]);  cos block likewise.
struct half_out_t {  // IEEE-half output tag (bf16_t is a plain unsigned short)
    unsigned short bits;
};
template <class TOut>  // bf16 / IEEE half (product path, training / inference) or fp32 (the fp32 checking mode, core/model/precise.py)
__global__ __launch_bounds__(256) void loftup_fourier_cn_kernel(const float* __restrict__ img,
                                                                 const float* __restrict__ mm /* [3][2] lo,hi */,
                                                                 const float* __restrict__ freqs,
                                                                 const float* __restrict__ bias_sin,
                                                                 const float* __restrict__ bias_cos,
                                                                 const float* __restrict__ gamma,
                                                                 const float* __restrict__ beta, TOut* __restrict__ out,
                                                                 int H, int W, int F, int ldo, float eps, long npix) {
    const int lane = threadIdx.x & 63;
    const long pix = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (pix >= npix) return;
    const long HW = (long)H * W;
    const long b = pix / HW, p = pix - b * HW;
    const int y = (int)(p / W), x = (int)(p - (long)y * W);
    float f5[5];
    f5[0] = linspace_pm1(y, H);
    f5[1] = linspace_pm1(x, W);
    float raw[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float lo = mm[2 * c], hi = mm[2 * c + 1];
        raw[c] = (img[((size_t)b * 3 + c) * HW + p] - lo) / fmaxf(hi - lo, 1e-4f) - 0.5f;
        f5[2 + c] = raw[c];
    }
    const int nsc = 5 * F, nfeat = 2 * nsc + 3;
    float v[4];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int ch = lane + 64 * i;
        float val = 0.f;
        if (ch < nsc) {
            const int f = ch / 5, m = ch - f * 5;
            // (product and sum rounded separately, as the reference's `feats * freqs` then `+ biases` are: at freq = e^10 one
            // ulp of the product is 2e-3 rad, so a fused multiply-add would move these phases by up to 1e-3)
            val = sinf(__fadd_rn(__fmul_rn(f5[m], freqs[f]), bias_sin[ch]));
        } else if (ch < 2 * nsc) {
            const int k = ch - nsc, f = k / 5, m = k - f * 5;
            val = cosf(__fadd_rn(__fmul_rn(f5[m], freqs[f]), bias_cos[k]));
        } else if (ch < nfeat) {
            val = raw[ch - 2 * nsc];
        }
        v[i] = val;
        sum += ch < nfeat ? val : 0.f;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    const float mean = sum / (float)nfeat;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float d = v[i] - mean;
        sq += (lane + 64 * i) < nfeat ? d * d : 0.f;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
    const float rstd = 1.0f / sqrtf(sq / (float)nfeat + eps);
    TOut* orow = out + pix * ldo;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int ch = lane + 64 * i;
        if (ch < ldo) {
            const float o = ch < nfeat ? (v[i] - mean) * rstd * gamma[ch] + beta[ch] : 0.f;
            if constexpr (sizeof(TOut) == 4) orow[ch] = o;
            else if constexpr (sizeof(TOut) == 2 && !__is_same(TOut, bf16_t)) orow[ch].bits = (unsigned short)(pack2h(o, 0.f) & 0xffffu);
            else orow[ch] = f2bf(o);
        }
    }
}

}  // namespace

extern "C" int isp_minmax_nchw_f32(const float* x, float* out_c2, float* workspace, int B, int C, long HW,
                                   void* stream) {
    ISP_CHECK_ARG(x && out_c2 && workspace && B > 0 && C > 0 && HW > 0 && B <= 65535 && C <= 65535);
    const int nblk = 64;  // workspace: C * B * 64 * 2 floats
    dim3 grid(nblk, C, B);
    minmax_partial_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(x, workspace, C, HW, nblk);
    minmax_final_kernel<<<C, 64, 0, (hipStream_t)stream>>>(workspace, out_c2, B * nblk);
    return isp_launch_status();
}

extern "C" int isp_loftup_fourier_cn(const float* image, const float* minmax_c2, const float* freqs,
                                     const float* bias_sin, const float* bias_cos, const float* gamma,
                                     const float* beta, void* out_bf16, int B, int H, int W, int n_freqs, int ldo,
                                     float eps, void* stream) {
    ISP_CHECK_ARG(image && minmax_c2 && freqs && bias_sin && bias_cos && gamma && beta && out_bf16);
    ISP_CHECK_ARG(B > 0 && H > 0 && W > 0 && n_freqs > 0 && 10 * n_freqs + 3 <= 256 && ldo >= 10 * n_freqs + 3 && ldo <= 256);
    const long npix = (long)B * H * W;
    loftup_fourier_cn_kernel<bf16_t><<<(unsigned)((npix + 3) / 4), 256, 0, (hipStream_t)stream>>>(
        image, minmax_c2, freqs, bias_sin, bias_cos, gamma, beta, (bf16_t*)out_bf16, H, W, n_freqs, ldo, eps, npix);
    return isp_launch_status();
}

extern "C" int isp_loftup_fourier_cn_f32(const float* image, const float* minmax_c2, const float* freqs, const float* bias_sin,
                                         const float* bias_cos, const float* gamma, const float* beta, float* out_f32, int B,
                                         int H, int W, int n_freqs, int ldo, float eps, void* stream) {
    ISP_CHECK_ARG(image && minmax_c2 && freqs && bias_sin && bias_cos && gamma && beta && out_f32);
    ISP_CHECK_ARG(B > 0 && H > 0 && W > 0 && n_freqs > 0 && 10 * n_freqs + 3 <= 256 && ldo >= 10 * n_freqs + 3 && ldo <= 256);
    const long npix = (long)B * H * W;
    loftup_fourier_cn_kernel<float><<<(unsigned)((npix + 3) / 4), 256, 0, (hipStream_t)stream>>>(
        image, minmax_c2, freqs, bias_sin, bias_cos, gamma, beta, out_f32, H, W, n_freqs, ldo, eps, npix);
    return isp_launch_status();
}

extern "C" int isp_loftup_fourier_cn_f16(const float* image, const float* minmax_c2, const float* freqs, const float* bias_sin,
                                         const float* bias_cos, const float* gamma, const float* beta, void* out_f16, int B,
                                         int H, int W, int n_freqs, int ldo, float eps, void* stream) {
    ISP_CHECK_ARG(image && minmax_c2 && freqs && bias_sin && bias_cos && gamma && beta && out_f16);
    ISP_CHECK_ARG(B > 0 && H > 0 && W > 0 && n_freqs > 0 && 10 * n_freqs + 3 <= 256 && ldo >= 10 * n_freqs + 3 && ldo <= 256);
    const long npix = (long)B * H * W;
    loftup_fourier_cn_kernel<half_out_t><<<(unsigned)((npix + 3) / 4), 256, 0, (hipStream_t)stream>>>(
        image, minmax_c2, freqs, bias_sin, bias_cos, gamma, beta, (half_out_t*)out_f16, H, W, n_freqs, ldo, eps, npix);
    return isp_launch_status();
}
