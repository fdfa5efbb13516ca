// Click -> disk / distance maps, image normalisation, and the patch-matrix builder that
// feeds the fused (image + click) patch-embed GEMM.  All HBM-bound elementwise kernels.
//
// Replaces: DistMaps.get_coord_features torch path (reference core/model/ops.py:35-77),
// get_dist_maps (core/utils/cython/_get_dist_maps.pyx:18-64, via round_clicks=1),
// BatchImageNormalize (ops.py:96-105), and the im2col side of both PatchEmbed convs
// (featurizers/utils/patch_embed.py:37-42, dinov2/layers/patch_embed.py:71-87).
#include "isp_common.h"

#define CM_MAX_CLICKS 64  // clicks staged in LDS per (sample, polarity) chunk

// One block = one row-tile of one (sample, polarity) map; a thread owns VEC consecutive
// columns.  Every op mirrors one fp32 torch op of ops.py:55-75 (no FMA contraction):
//   t  = p * scale                (points * spatial_scale)
//   d  = coord + (-t)             (coords.add_(-add_xy))
//   d /= radius*scale             (only when !disks)
//   d2 = dr*dr + dc*dc            (mul_, row + col)
//   min over clicks, 1e6 when the polarity has no valid click
//   disks: d2 <= (radius*scale)^2 ; else tanh(2*sqrt(d2))
template <int VEC>
__global__ __launch_bounds__(256) void click_maps_kernel(const float* __restrict__ points, float* __restrict__ out,
                                                          int P, int H, int W, float scale, float denom, float thr,
                                                          int use_disks, int round_clicks) {
    __shared__ float s_r[CM_MAX_CLICKS], s_c[CM_MAX_CLICKS];
    __shared__ int s_n;
    const int bp = blockIdx.y;  // b*2 + polarity
    const float* pts = points + (size_t)bp * P * 3;
    const int wv = (W + VEC - 1) / VEC;
    const long total = (long)H * wv;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const bool active = idx < total;
    const int row = active ? (int)(idx / wv) : 0;
    const int col0 = active ? (int)(idx % wv) * VEC : 0;
    const float fr = (float)row;
    float best[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) best[v] = 1e6f;

    for (int base = 0; base < P; base += CM_MAX_CLICKS) {
        __syncthreads();
        if (threadIdx.x == 0) s_n = 0;
        __syncthreads();
        const int k = base + threadIdx.x;
        if (threadIdx.x < CM_MAX_CLICKS && k < P) {
            float pr = pts[3 * k + 0], pc = pts[3 * k + 1];
            bool valid;
            if (round_clicks) {  // Cython path: round-half-even, skip when rounded row < 0 (pyx:31-32)
                pr = rintf(pr);
                pc = rintf(pc);
                valid = pr >= 0.f;
            } else {  // torch path: invalid iff max(row, col) < 0 (ops.py:40)
                valid = fmaxf(pr, pc) >= 0.f;
            }
            if (valid) {
                const int slot = atomicAdd(&s_n, 1);
                s_r[slot] = __fmul_rn(pr, scale);
                s_c[slot] = __fmul_rn(pc, scale);
            }
        }
        __syncthreads();
        const int n = s_n;
        for (int i = 0; i < n; ++i) {
            float dr = __fadd_rn(fr, -s_r[i]);
            if (!use_disks) dr = __fdiv_rn(dr, denom);
            const float dr2 = __fmul_rn(dr, dr);
            const float tc = s_c[i];
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                float dc = __fadd_rn((float)(col0 + v), -tc);
                if (!use_disks) dc = __fdiv_rn(dc, denom);
                best[v] = fminf(best[v], __fadd_rn(dr2, __fmul_rn(dc, dc)));
            }
        }
    }
    if (!active) return;
    float res[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v)
        res[v] = use_disks ? (best[v] <= thr ? 1.f : 0.f) : tanhf(__fmul_rn(__fsqrt_rn(best[v]), 2.f));
    float* o = out + ((size_t)bp * H + row) * W + col0;
    if (VEC == 4) {
        *reinterpret_cast<float4*>(o) = make_float4(res[0], res[1], res[2], res[3]);
    } else {
#pragma unroll
        for (int v = 0; v < VEC; ++v)
            if (col0 + v < W) o[v] = res[v];
    }
}

extern "C" int isp_click_maps_fwd(const float* points, float* out, int B, int P, int H, int W, float norm_radius,
                                  float spatial_scale, int use_disks, int round_clicks, void* stream) {
    ISP_CHECK_ARG(points && out && B > 0 && P > 0 && H > 0 && W > 0 && norm_radius > 0.f && spatial_scale > 0.f);
    ISP_CHECK_ARG(B * 2 <= 65535);
    // python-double products rounded to fp32, as torch does with scalar operands
    const float denom = (float)((double)norm_radius * (double)spatial_scale);
    const double rs = (double)norm_radius * (double)spatial_scale;
    const float thr = (float)(rs * rs);
    hipStream_t s = (hipStream_t)stream;
    if (W % 4 == 0) {
        const long total = (long)H * (W / 4);
        dim3 grid((unsigned)((total + 255) / 256), B * 2);
        click_maps_kernel<4><<<grid, 256, 0, s>>>(points, out, P, H, W, spatial_scale, denom, thr, use_disks, round_clicks);
    } else {
        const long total = (long)H * W;
        dim3 grid((unsigned)((total + 255) / 256), B * 2);
        click_maps_kernel<1><<<grid, 256, 0, s>>>(points, out, P, H, W, spatial_scale, denom, thr, use_disks, round_clicks);
    }
    return isp_launch_status();
}

// ---------------------------------------------------------------------------------------
// (x - mean) / std on the first 3 channels of an NCHW fp32 image with in_ch (3 or 4)
// channels; optionally copies channel 3 (previous mask) out.  ops.py:101-105,
// iseg_base_model.py:91-98.
__global__ __launch_bounds__(256) void normalize_kernel(const float* __restrict__ img, float* __restrict__ out,
                                                         float* __restrict__ prev, int in_ch, long hw, float m0,
                                                         float m1, float m2, float s0, float s1, float s2) {
    const long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i >= hw) return;
    const int b = blockIdx.y;
    const float* src = img + (size_t)b * in_ch * hw;
    const float mean[3] = {m0, m1, m2}, sd[3] = {s0, s1, s2};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float4 v = *reinterpret_cast<const float4*>(src + c * hw + i);
        float4 r;
        r.x = __fdiv_rn(__fadd_rn(v.x, -mean[c]), sd[c]);
        r.y = __fdiv_rn(__fadd_rn(v.y, -mean[c]), sd[c]);
        r.z = __fdiv_rn(__fadd_rn(v.z, -mean[c]), sd[c]);
        r.w = __fdiv_rn(__fadd_rn(v.w, -mean[c]), sd[c]);
        *reinterpret_cast<float4*>(out + ((size_t)b * 3 + c) * hw + i) = r;
    }
    if (prev && in_ch == 4)
        *reinterpret_cast<float4*>(prev + (size_t)b * hw + i) = *reinterpret_cast<const float4*>(src + 3 * hw + i);
}

extern "C" int isp_normalize_fwd(const float* image, float* out, float* prev_mask, int B, int in_ch, int H, int W,
                                 const float* mean3, const float* std3, void* stream) {
    ISP_CHECK_ARG(image && out && mean3 && std3 && B > 0 && (in_ch == 3 || in_ch == 4) && H > 0 && W > 0);
    const long hw = (long)H * W;
    ISP_CHECK_ARG(hw % 4 == 0 && B <= 65535);
    dim3 grid((unsigned)((hw / 4 + 255) / 256), B);
    normalize_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(image, out, prev_mask, in_ch, hw, mean3[0], mean3[1],
                                                            mean3[2], std3[0], std3[1], std3[2]);
    return isp_launch_status();
}

// ---------------------------------------------------------------------------------------
// Patch matrix for the fused patch-embed GEMM.  Row t of sample b holds, as bf16,
//   [ img(c=0..n_img-1, i, j) | coord(c=0..nc-1, i, j) | 0 padding ]   (c-major, then i, then j:
// the flattening order of a Conv2d weight [D, C, p, p]), so that
//   tokens = A . [W_img | W_click]^T  reproduces  patch_embed(image) + embed_coords(coord)
// (DINOv2.py:518-523) in ONE GEMM.  coord channels come from up to two NCHW fp32 tensors
// (prev mask [B,1,H,W] then click maps [B,2,H,W]; either may be null).
__global__ __launch_bounds__(256) void patchify_kernel(const float* __restrict__ img, const float* __restrict__ prev,
                                                        const float* __restrict__ maps, bf16_t* __restrict__ A, int H,
                                                        int W, int p, int gw, int hw_tokens, int n_img, int n_prev,
                                                        int n_maps, int Kpad) {
    // one block per token; threads sweep the K axis
    const int tok = blockIdx.x, b = blockIdx.y;
    const int ty = tok / gw, tx = tok % gw;
    const int pp = p * p;
    const int kimg = n_img * pp, kall = (n_img + n_prev + n_maps) * pp;
    bf16_t* row = A + ((size_t)b * hw_tokens + tok) * Kpad;
    const size_t plane = (size_t)H * W;
    for (int k = threadIdx.x; k < Kpad; k += blockDim.x) {
        float v = 0.f;
        if (k < kall) {
            const int c = k / pp, r = k % pp, i = r / p, j = r % p;
            const size_t off = (size_t)(ty * p + i) * W + tx * p + j;
            if (k < kimg)
                v = img[((size_t)b * n_img + c) * plane + off];
            else if (c - n_img < n_prev)
                v = prev[((size_t)b * n_prev + (c - n_img)) * plane + off];
            else
                v = maps[((size_t)b * n_maps + (c - n_img - n_prev)) * plane + off];
        }
        row[k] = f2bf(v);
    }
}

// The same matrix for a whole ROW of tokens per block (the bench shapes: p = 14, W = 448 -> 32 tokens): each channel's
// p x W strip is read as whole image rows (coalesced, through LDS) and leaves as p*p-element runs of the token rows in
// 8-byte stores.  One block per token gathered 56-byte runs with two integer divisions per element: 136 us at batch 32
// x 448^2 for 234 MB of traffic.  Requires p even (8-byte alignment of the runs) and Kpad % 4 == 0.
constexpr int PATCH_STRIP_MAX = 16384;  // floats of LDS (64 KiB): p * W <= 16384 covers 14 x 896
__global__ __launch_bounds__(256) void patchify_rows_kernel(const float* __restrict__ img, const float* __restrict__ prev,
                                                             const float* __restrict__ maps, bf16_t* __restrict__ A, int H,
                                                             int W, int p, int gw, int hw_tokens, int n_img, int n_prev,
                                                             int n_maps, int Kpad) {
    __shared__ float strip[PATCH_STRIP_MAX];
    const int ty = blockIdx.x, b = blockIdx.y;
    const int pp = p * p, nch = n_img + n_prev + n_maps, kall = nch * pp;
    const size_t plane = (size_t)H * W;
    bf16_t* rows = A + ((size_t)b * hw_tokens + (size_t)ty * gw) * Kpad;
    for (int c = 0; c < nch; ++c) {
        const float* src = c < n_img ? img + ((size_t)b * n_img + c) * plane
                                     : (c - n_img < n_prev ? prev + ((size_t)b * n_prev + (c - n_img)) * plane
                                                           : maps + ((size_t)b * n_maps + (c - n_img - n_prev)) * plane);
        src += (size_t)ty * p * W;
        __syncthreads();  // the previous channel's strip has been written out
        for (int i = threadIdx.x * 4; i < p * W; i += 1024)  // (W % 4 == 0: checked by the launcher)
            *reinterpret_cast<float4*>(strip + i) = *reinterpret_cast<const float4*>(src + i);
        __syncthreads();
        // token tx, element quad q of its p*p run: r = 4q .. 4q+3 -> (i, j) = (r / p, r % p); j + 3 < p because p % 2 == 0
        // does not guarantee it -- take the four elements one by one
        const int quads = pp / 4;
        for (int w = threadIdx.x; w < gw * quads; w += 256) {
            const int tx = w / quads, q = w - tx * quads;
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int r = 4 * q + e, i = r / p, j = r - i * p;
                v[e] = strip[i * W + tx * p + j];
            }
            *reinterpret_cast<uint2*>(rows + (size_t)tx * Kpad + c * pp + 4 * q) =
                make_uint2((unsigned)f2bf(v[0]) | ((unsigned)f2bf(v[1]) << 16), (unsigned)f2bf(v[2]) | ((unsigned)f2bf(v[3]) << 16));
        }
    }
    // zero padding behind the last channel
    for (int w = threadIdx.x; w < gw * (Kpad - kall); w += 256) {
        const int tx = w / (Kpad - kall), k = w - tx * (Kpad - kall);
        rows[(size_t)tx * Kpad + kall + k] = 0;
    }
}

extern "C" int isp_patchify_fwd(const float* image, const float* prev_mask, const float* click_maps, void* A_bf16,
                                int B, int H, int W, int patch, int n_img, int n_prev, int n_maps, int Kpad,
                                void* stream) {
    ISP_CHECK_ARG(A_bf16 && B > 0 && patch > 0 && H > 0 && W > 0 && H % patch == 0 && W % patch == 0);
    ISP_CHECK_ARG(n_img >= 0 && n_prev >= 0 && n_maps >= 0 && n_img + n_prev + n_maps > 0);
    ISP_CHECK_ARG((n_img == 0 || image) && (n_prev == 0 || prev_mask) && (n_maps == 0 || click_maps));
    ISP_CHECK_ARG(Kpad >= (n_img + n_prev + n_maps) * patch * patch && Kpad % 8 == 0 && B <= 65535);
    const int gh = H / patch, gw = W / patch;
    if ((patch * patch) % 4 == 0 && W % 4 == 0 && Kpad % 4 == 0 && patch * W <= PATCH_STRIP_MAX && gh <= 65535) {
        dim3 grid_rows(gh, B);
        patchify_rows_kernel<<<grid_rows, 256, 0, (hipStream_t)stream>>>(image, prev_mask, click_maps, (bf16_t*)A_bf16, H, W, patch,
                                                                         gw, gh * gw, n_img, n_prev, n_maps, Kpad);
        return isp_launch_status();
    }
    dim3 grid(gh * gw, B);
    patchify_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(image, prev_mask, click_maps, (bf16_t*)A_bf16, H, W, patch,
                                                           gw, gh * gw, n_img, n_prev, n_maps, Kpad);
    return isp_launch_status();
}
