// First head convolution taken THROUGH the bilinear resize (gfx950).
//
//   reference: F.interpolate(x, size=(H, W), mode="bilinear", align_corners=True)   core/model/iseg_probe_model.py:120-129
//              (or the bilinear upsampler plugin itself, upsamplers/basic_upsamplers.py:28-33)
//              followed by ConvModule(3x3, pad 1, bias) + ReLU                        core/model/heads/conv_heads.py:59-73
//
// The resized map Y[p] = sum_{q in 2x2(p)} a(p, q) X[q] is a blend of a (H/h)^2 times smaller map, and the convolution is
// linear in it, so
//     out[p][n] = act(b[n] + sum_t [p + t inside] sum_c W_t[n][c] Y[p + t][c])
//               = act(b[n] + sum_t [p + t inside] sum_{q in 2x2(p + t)} a(p + t, q) Z_t[q][n]),     Z_t = X W_t^T  (low resolution).
// Z = X [B h w, C] x [W_0 .. W_8]^T is ONE dense GEMM at low resolution ([B h w, 9 N], column t N + n; isp_gemm_f16) and
// the kernel below is the blend: 36 multiply-adds per output value instead of 9 C (= 3 456 at C = 384, 9 216 at C = 1024),
// and the [B, H, W, C] map (4.9 GB at batch 32 x 448^2 x 384, 1.6 GB per image at 896^2 x 1024) never exists.
//
// Kernel: a workgroup owns a 16 x 16 patch of output pixels, one pixel per thread.  The source pixels its 18 x 18 tap
// neighbourhood touches (<= FMAX x FMAX, checked by the launcher with the kernel's own fp32 coordinate arithmetic) are staged
// per channel block in LDS as [q][tap][CB channels]; a thread keeps 12 row / column weights (tap validity folded in as zeros,
// coordinates clamped so the addresses stay inside the staged footprint) and 12 row / column offsets in registers, reads
// 16-byte channel groups (neighbouring pixels mostly share their corners: LDS broadcasts) and accumulates in fp32
// (v_fma_mix_f32 takes the half operand as is).  Per channel block and CU the LDS port and the vector pipe are equally
// loaded (4.6 k cycles each per 256 pixels x 64 channels); HBM sees the output map once, Z stays in L2 / Infinity Cache.
#include "isp_common.h"

namespace {

constexpr int FMAX = 5;      // staged source footprint per axis
constexpr int TPX = 16;      // output tile edge

template <typename ZT>
struct ZTraits;
template <>
struct ZTraits<_Float16> {
    static constexpr int CB = 64;  // channels per block
};
template <>
struct ZTraits<float> {
    static constexpr int CB = 32;
};

template <int OUT>
__device__ __forceinline__ void store8(void* out, size_t idx, const float* v) {
    if constexpr (OUT == ISP_F32) {
        float4* o = reinterpret_cast<float4*>(reinterpret_cast<float*>(out) + idx);
        o[0] = make_float4(v[0], v[1], v[2], v[3]);
        o[1] = make_float4(v[4], v[5], v[6], v[7]);
    } else if constexpr (OUT == ISP_F16) {
        *reinterpret_cast<uint4*>(reinterpret_cast<unsigned short*>(out) + idx) =
            make_uint4(pack2h_sat(v[0], v[1]), pack2h_sat(v[2], v[3]), pack2h_sat(v[4], v[5]), pack2h_sat(v[6], v[7]));
    } else {
        *reinterpret_cast<uint4*>(reinterpret_cast<unsigned short*>(out) + idx) =
            make_uint4(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]), pack2bf(v[4], v[5]), pack2bf(v[6], v[7]));
    }
}

// src coordinate of an output coordinate, exactly as the resize kernels (and torch) compute it: fp32 product, truncation
__device__ __host__ __forceinline__ void src_coord(int d, float s, int n_in, int& i0, int& i1, float& l) {
    const float f = s * (float)d;
    i0 = (int)f;
    i1 = i0 + 1 < n_in ? i0 + 1 : n_in - 1;
    l = f - (float)i0;
}

template <typename ZT, int OUT, bool RELU>
__global__ __launch_bounds__(256, 3) void conv_bilinear_blend_kernel(const ZT* __restrict__ z, const float* __restrict__ bias,
                                                                  void* __restrict__ out, int h, int w, int H, int W, int N,
                                                                  float sy, float sx) {
    constexpr int CB = ZTraits<ZT>::CB;
    constexpr int QROW = 9 * CB * (int)sizeof(ZT);  // bytes of one staged source pixel: [tap][CB]
    constexpr int QPITCH = QROW + 16;               // +16: consecutive source pixels start 4 banks apart (mod 64: 36, 8, 44, ...)
    constexpr int PIECES = QROW / 16;               // 16-byte pieces per source pixel
    constexpr int TAPB = CB * (int)sizeof(ZT);      // bytes per tap
    __shared__ __attribute__((aligned(16))) char zs[FMAX * FMAX * QPITCH];

    const int tid = threadIdx.x;
    const int b = blockIdx.z;
    const int Y0 = blockIdx.y * TPX, X0 = blockIdx.x * TPX;
    // footprint of the tile's clipped 18 x 18 neighbourhood (block-uniform)
    int qy_lo, qx_lo, fy, fx;
    {
        int i0, i1;
        float l;
        src_coord(max(Y0 - 1, 0), sy, h, i0, i1, l);
        qy_lo = i0;
        src_coord(min(Y0 + TPX, H - 1), sy, h, i0, i1, l);
        fy = i1 - qy_lo + 1;
        src_coord(max(X0 - 1, 0), sx, w, i0, i1, l);
        qx_lo = i0;
        src_coord(min(X0 + TPX, W - 1), sx, w, i0, i1, l);
        fx = i1 - qx_lo + 1;
    }
    const int py = tid >> 4, px = tid & 15;
    const int Y = Y0 + py, X = X0 + px;
    const bool live = Y < H && X < W;
    // per tap row / column: two LDS byte offsets and two weights (zero when the tap leaves the image)
    int roff[3][2], coff[3][2];
    float wy[3][2], wx[3][2];
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        int i0, i1;
        float l;
        const int yy = Y + t - 1, xx = X + t - 1;
        const bool vy = yy >= 0 && yy < H, vx = xx >= 0 && xx < W;
        src_coord(min(max(yy, 0), H - 1), sy, h, i0, i1, l);
        i0 = min(max(i0 - qy_lo, 0), fy - 1), i1 = min(max(i1 - qy_lo, 0), fy - 1);  // (dead threads of a partial tile stay in range)
        roff[t][0] = i0 * fx * QPITCH, roff[t][1] = i1 * fx * QPITCH;
        wy[t][0] = vy ? 1.f - l : 0.f, wy[t][1] = vy ? l : 0.f;
        src_coord(min(max(xx, 0), W - 1), sx, w, i0, i1, l);
        i0 = min(max(i0 - qx_lo, 0), fx - 1), i1 = min(max(i1 - qx_lo, 0), fx - 1);
        coff[t][0] = i0 * QPITCH, coff[t][1] = i1 * QPITCH;
        wx[t][0] = vx ? 1.f - l : 0.f, wx[t][1] = vx ? l : 0.f;
    }
    const size_t zrow = (size_t)9 * N;  // elements per source pixel in Z
    const ZT* zb = z + ((size_t)b * h * w) * zrow;
    const int nq = fy * fx;
    const size_t opix = ((size_t)b * H + Y) * W + X;

    for (int n0 = 0; n0 < N; n0 += CB) {
        __syncthreads();  // previous block's reads are done
        // the 36 corner addresses and weights are re-derived per channel block from these 24 values (hoisted out of the
        // loop they cost 72 registers = one wave per SIMD)
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                asm volatile("" : "+v"(roff[t][c]));
                asm volatile("" : "+v"(coff[t][c]));
                asm volatile("" : "+v"(wy[t][c]));
                asm volatile("" : "+v"(wx[t][c]));
            }
        // ---- stage [q][tap][CB]: piece i of source pixel q = 16 bytes of tap i / (PIECES / 9)
        for (int i = tid; i < nq * PIECES; i += 256) {
            const int q = i / PIECES, pc = i - q * PIECES;
            const int t = pc / (PIECES / 9), r = pc - t * (PIECES / 9);
            const int qy = qy_lo + q / fx, qx = qx_lo + q % fx;
            const ZT* src = zb + ((size_t)qy * w + qx) * zrow + (size_t)t * N + n0 + r * (16 / (int)sizeof(ZT));
            *reinterpret_cast<uint4*>(zs + q * QPITCH + pc * 16) = *reinterpret_cast<const uint4*>(src);
        }
        __syncthreads();
        float acc[CB];
#pragma unroll
        for (int c = 0; c < CB; ++c) acc[c] = 0.f;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const char* p = zs + roff[t / 3][c >> 1] + coff[t % 3][c & 1] + t * TAPB;
                const float wv = wy[t / 3][c >> 1] * wx[t % 3][c & 1];
                if constexpr (sizeof(ZT) == 2) {
#pragma unroll
                    for (int g = 0; g < CB / 8; ++g) {
                        const f16x8_t v = *reinterpret_cast<const f16x8_t*>(p + g * 16);
#pragma unroll
                        for (int e = 0; e < 8; ++e) acc[g * 8 + e] = fmaf((float)v[e], wv, acc[g * 8 + e]);
                    }
                } else {
#pragma unroll
                    for (int g = 0; g < CB / 4; ++g) {
                        const float4 v = *reinterpret_cast<const float4*>(p + g * 16);
                        acc[g * 4 + 0] = fmaf(v.x, wv, acc[g * 4 + 0]);
                        acc[g * 4 + 1] = fmaf(v.y, wv, acc[g * 4 + 1]);
                        acc[g * 4 + 2] = fmaf(v.z, wv, acc[g * 4 + 2]);
                        acc[g * 4 + 3] = fmaf(v.w, wv, acc[g * 4 + 3]);
                    }
                }
                // keep the groups in order: left alone all 288 fragment reads are hoisted in front of the multiply-adds (724
                // spilled registers); the accumulators pass through an opaque asm so that a group's arithmetic cannot sink
#pragma unroll
                for (int e = 0; e < CB; ++e) asm volatile("" : "+v"(acc[e]));
            }
        }
        if (live) {
#pragma unroll
            for (int g = 0; g < CB / 8; ++g) {
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    v[e] = acc[g * 8 + e] + (bias ? bias[n0 + g * 8 + e] : 0.f);
                    if (RELU) v[e] = fmaxf(v[e], 0.f);
                }
                store8<OUT>(out, opix * N + n0 + g * 8, v);
            }
        }
    }
}

// ---- two-phase form of the same tile (the default): the blend is separable, a(p + t, q) = ay(Y + ty, qy) ax(X + tx, qx), so a
// tile first combines the tap planes ALONG X at the source rows it touches,
//     U[ty][qy][X][n] = sum_tx [X + tx inside] sum_{qx in 2(X + tx)} ax(X + tx, qx) Z[qy][qx][(ty, tx)][n]      (6 multiply-adds),
// into LDS (3 x fy x 16 entries per channel, one entry per thread), and then blends ALONG Y per output pixel,
//     out[Y][X][n] = act(b[n] + sum_ty [Y + ty inside] sum_{qy in 2(Y + ty)} ay(Y + ty, qy) U[ty][qy][X][n])       (6 multiply-adds):
// 12 multiply-adds through 12 LDS reads per output value instead of 36 through 36 (a tile's U entries are shared by its 16
// rows), i.e. 3.8x less vector work and 2.3x less LDS traffic per tile than the one-phase kernel above -- which leaves the
// kernel on the HBM write of the output map.  U is kept in the tap planes' precision (half on the product route, where it is
// one more 2^-12 rounding; fp32 on the checking route).
template <typename ZT, int OUT, bool RELU>
__global__ __launch_bounds__(256, 3) void conv_bilinear_blend2_kernel(const ZT* __restrict__ z, const float* __restrict__ bias,
                                                                      void* __restrict__ out, int h, int w, int H, int W, int N,
                                                                      float sy, float sx, int zs_bytes) {
    constexpr int CB = ZTraits<ZT>::CB;
    constexpr int ES = (int)sizeof(ZT);
    constexpr int QROW = 9 * CB * ES, QPITCH = QROW + 16, PIECES = QROW / 16, TAPB = CB * ES;
    constexpr int UPITCH = CB * ES + 16;  // bytes per (ty, qy, X) entry: 16 consecutive X start 4 banks apart (x36 / x68 dwords mod 64)
    constexpr int VEC = 16 / ES;          // channels per 16-byte read
    // dynamic LDS sized by the launcher for the largest footprint any tile of this geometry has (3 x 3 source pixels at x14:
    // 31 KiB, five workgroups per CU; the 5 x 5 worst case would be 64 KiB)
    extern __shared__ __attribute__((aligned(16))) char lds[];
    char* const zs = lds;
    char* const us = lds + zs_bytes;

    const int tid = threadIdx.x;
    const int b = blockIdx.z;
    const int Y0 = blockIdx.y * TPX, X0 = blockIdx.x * TPX;
    int qy_lo, qx_lo, fy, fx;
    {
        int i0, i1;
        float l;
        src_coord(max(Y0 - 1, 0), sy, h, i0, i1, l);
        qy_lo = i0;
        src_coord(min(Y0 + TPX, H - 1), sy, h, i0, i1, l);
        fy = i1 - qy_lo + 1;
        src_coord(max(X0 - 1, 0), sx, w, i0, i1, l);
        qx_lo = i0;
        src_coord(min(X0 + TPX, W - 1), sx, w, i0, i1, l);
        fx = i1 - qx_lo + 1;
    }
    // ---- phase-A role: entry (ty, qy, X) of U
    const bool a_live = tid < 3 * fy * TPX;
    const int a_ty = a_live ? tid / (fy * TPX) : 0;
    const int a_r = tid - a_ty * fy * TPX;
    const int a_qy = a_live ? a_r / TPX : 0, a_x = a_r % TPX;
    int zoff[3][2];
    float wx[3][2];
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        int i0, i1;
        float l;
        const int xx = X0 + a_x + t - 1;
        const bool vx = xx >= 0 && xx < W;
        src_coord(min(max(xx, 0), W - 1), sx, w, i0, i1, l);
        i0 = min(max(i0 - qx_lo, 0), fx - 1), i1 = min(max(i1 - qx_lo, 0), fx - 1);
        zoff[t][0] = (a_qy * fx + i0) * QPITCH + (a_ty * 3 + t) * TAPB;
        zoff[t][1] = (a_qy * fx + i1) * QPITCH + (a_ty * 3 + t) * TAPB;
        wx[t][0] = vx ? 1.f - l : 0.f, wx[t][1] = vx ? l : 0.f;
    }
    const int uw_off = ((a_ty * fy + a_qy) * TPX + a_x) * UPITCH;
    // ---- phase-B role: output pixel (Y, X)
    const int py = tid >> 4, px = tid & 15;
    const int Y = Y0 + py, X = X0 + px;
    const bool live = Y < H && X < W;
    int uoff[3][2];
    float wy[3][2];
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        int i0, i1;
        float l;
        const int yy = Y + t - 1;
        const bool vy = yy >= 0 && yy < H;
        src_coord(min(max(yy, 0), H - 1), sy, h, i0, i1, l);
        i0 = min(max(i0 - qy_lo, 0), fy - 1), i1 = min(max(i1 - qy_lo, 0), fy - 1);
        uoff[t][0] = ((t * fy + i0) * TPX + px) * UPITCH, uoff[t][1] = ((t * fy + i1) * TPX + px) * UPITCH;
        wy[t][0] = vy ? 1.f - l : 0.f, wy[t][1] = vy ? l : 0.f;
    }
    const size_t zrow = (size_t)9 * N;
    const ZT* zb = z + ((size_t)b * h * w) * zrow;
    const int nq = fy * fx;
    const size_t opix = ((size_t)b * H + Y) * W + X;

    auto accumulate = [&](float (&acc)[CB], const char* p, float wv) {
        if constexpr (ES == 2) {
#pragma unroll
            for (int g = 0; g < CB / 8; ++g) {
                const f16x8_t v = *reinterpret_cast<const f16x8_t*>(p + g * 16);
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[g * 8 + e] = fmaf((float)v[e], wv, acc[g * 8 + e]);
            }
        } else {
#pragma unroll
            for (int g = 0; g < CB / 4; ++g) {
                const float4 v = *reinterpret_cast<const float4*>(p + g * 16);
                acc[g * 4 + 0] = fmaf(v.x, wv, acc[g * 4 + 0]);
                acc[g * 4 + 1] = fmaf(v.y, wv, acc[g * 4 + 1]);
                acc[g * 4 + 2] = fmaf(v.z, wv, acc[g * 4 + 2]);
                acc[g * 4 + 3] = fmaf(v.w, wv, acc[g * 4 + 3]);
            }
        }
#pragma unroll
        for (int e = 0; e < CB; ++e) asm volatile("" : "+v"(acc[e]));  // keep the six groups in order (see the kernel above)
    };

    for (int n0 = 0; n0 < N; n0 += CB) {
        __syncthreads();  // the previous block's phase B is done with us, its phase A with zs
        for (int i = tid; i < nq * PIECES; i += 256) {
            const int q = i / PIECES, pc = i - q * PIECES;
            const int t = pc / (PIECES / 9), r = pc - t * (PIECES / 9);
            const int qy = qy_lo + q / fx, qx = qx_lo + q % fx;
            const ZT* src = zb + ((size_t)qy * w + qx) * zrow + (size_t)t * N + n0 + r * VEC;
            *reinterpret_cast<uint4*>(zs + q * QPITCH + pc * 16) = *reinterpret_cast<const uint4*>(src);
        }
        __syncthreads();
        if (a_live) {  // phase A
            float acc[CB];
#pragma unroll
            for (int c = 0; c < CB; ++c) acc[c] = 0.f;
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
                for (int j = 0; j < 2; ++j) accumulate(acc, zs + zoff[t][j], wx[t][j]);
            char* up = us + uw_off;
            if constexpr (ES == 2) {
#pragma unroll
                for (int g = 0; g < CB / 8; ++g)
                    *reinterpret_cast<uint4*>(up + g * 16) = make_uint4(pack2h_sat(acc[g * 8], acc[g * 8 + 1]), pack2h_sat(acc[g * 8 + 2], acc[g * 8 + 3]),
                                                                        pack2h_sat(acc[g * 8 + 4], acc[g * 8 + 5]), pack2h_sat(acc[g * 8 + 6], acc[g * 8 + 7]));
            } else {
#pragma unroll
                for (int g = 0; g < CB / 4; ++g)
                    *reinterpret_cast<float4*>(up + g * 16) = make_float4(acc[g * 4], acc[g * 4 + 1], acc[g * 4 + 2], acc[g * 4 + 3]);
            }
        }
        __syncthreads();
        {  // phase B
            float acc[CB];
#pragma unroll
            for (int c = 0; c < CB; ++c) acc[c] = 0.f;
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
                for (int i = 0; i < 2; ++i) accumulate(acc, us + uoff[t][i], wy[t][i]);
            if (live) {
#pragma unroll
                for (int g = 0; g < CB / 8; ++g) {
                    float v[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        v[e] = acc[g * 8 + e] + (bias ? bias[n0 + g * 8 + e] : 0.f);
                        if (RELU) v[e] = fmaxf(v[e], 0.f);
                    }
                    store8<OUT>(out, opix * N + n0 + g * 8, v);
                }
            }
        }
    }
}

// largest per-axis footprint over the tiles, with the kernel's own arithmetic
int max_footprint(int n_in, int n_out, float s) {
    int worst = 0;
    for (int t0 = 0; t0 < n_out; t0 += TPX) {
        int lo, hi, i1;
        float l;
        src_coord(t0 - 1 > 0 ? t0 - 1 : 0, s, n_in, lo, i1, l);
        src_coord(t0 + TPX < n_out - 1 ? t0 + TPX : n_out - 1, s, n_in, hi, i1, l);
        if (i1 - lo + 1 > worst) worst = i1 - lo + 1;
    }
    return worst;
}

template <typename ZT, int OUT>
int launch(const void* z, const float* bias, void* out, int B, int h, int w, int H, int W, int N, int relu, hipStream_t s) {
    const float sy = H > 1 ? (float)(h - 1) / (float)(H - 1) : 0.f;
    const float sx = W > 1 ? (float)(w - 1) / (float)(W - 1) : 0.f;
    const dim3 grid((W + TPX - 1) / TPX, (H + TPX - 1) / TPX, B);
    static const bool one_phase = [] { const char* e = getenv("ISEGPROBE_BLEND_ONE_PHASE"); return e && e[0] == '1'; }();  // A/B switch
    if (one_phase) {
        if (relu)
            conv_bilinear_blend_kernel<ZT, OUT, true><<<grid, 256, 0, s>>>((const ZT*)z, bias, out, h, w, H, W, N, sy, sx);
        else
            conv_bilinear_blend_kernel<ZT, OUT, false><<<grid, 256, 0, s>>>((const ZT*)z, bias, out, h, w, H, W, N, sy, sx);
    } else {
        constexpr int CB = ZTraits<ZT>::CB, ES = (int)sizeof(ZT);
        const int FY = max_footprint(h, H, sy), FX = max_footprint(w, W, sx);
        const int zs_bytes = FY * FX * (9 * CB * ES + 16);
        const int lds = zs_bytes + 3 * FY * TPX * (CB * ES + 16);  // <= 63 760 B at the 5 x 5 limit: inside the default 64 KiB
        if (relu)
            conv_bilinear_blend2_kernel<ZT, OUT, true><<<grid, 256, lds, s>>>((const ZT*)z, bias, out, h, w, H, W, N, sy, sx, zs_bytes);
        else
            conv_bilinear_blend2_kernel<ZT, OUT, false><<<grid, 256, lds, s>>>((const ZT*)z, bias, out, h, w, H, W, N, sy, sx, zs_bytes);
    }
    return isp_launch_status();
}

}  // namespace

extern "C" int isp_conv3x3_of_bilinear_supported(int h, int w, int H, int W, int N, int z_dtype) {
    if (h <= 0 || w <= 0 || H <= 0 || W <= 0 || N <= 0) return 0;
    if (z_dtype != ISP_F16 && z_dtype != ISP_F32) return 0;
    if (N % (z_dtype == ISP_F16 ? 64 : 32)) return 0;
    const float sy = H > 1 ? (float)(h - 1) / (float)(H - 1) : 0.f;
    const float sx = W > 1 ? (float)(w - 1) / (float)(W - 1) : 0.f;
    return max_footprint(h, H, sy) <= FMAX && max_footprint(w, W, sx) <= FMAX;
}

extern "C" int isp_conv3x3_of_bilinear_blend(const void* z, int z_dtype, const float* bias, void* out, int out_dtype, int B,
                                             int h, int w, int H, int W, int N, int relu, void* stream) {
    ISP_CHECK_ARG(z && out && B > 0 && B <= 65535 && h > 0 && w > 0 && H > 0 && W > 0 && N > 0);
    ISP_CHECK_ARG(((uintptr_t)z & 15) == 0 && ((uintptr_t)out & 15) == 0);
    if (!isp_conv3x3_of_bilinear_supported(h, w, H, W, N, z_dtype)) return ISP_ERR_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    if (z_dtype == ISP_F16) {
        if (out_dtype == ISP_F16) return launch<_Float16, ISP_F16>(z, bias, out, B, h, w, H, W, N, relu, s);
        if (out_dtype == ISP_BF16) return launch<_Float16, ISP_BF16>(z, bias, out, B, h, w, H, W, N, relu, s);
        if (out_dtype == ISP_F32) return launch<_Float16, ISP_F32>(z, bias, out, B, h, w, H, W, N, relu, s);
    } else {
        if (out_dtype == ISP_F32) return launch<float, ISP_F32>(z, bias, out, B, h, w, H, W, N, relu, s);
    }
    return ISP_ERR_UNSUPPORTED;
}
