// Backward kernels of the trainable tail of the path (seg head, resize, click patch-embed) and the row /
// elementwise pieces of the frozen-trunk backward: what loss.backward() (reference core/training/trainer.py:224 on
// models/*/patch-embed_*.py) needs besides the GEMM / conv engine (gemm.hip), the fused nine-tap conv weight
// gradient (conv_wgrad.hip) and the attention backward (attention_bwd.hip).
//
//   isp_tn_gemm_bf16_atomic   Out[n][j] += sum_m P[m][n] * Q[m'][j]   (weight gradients: both
//                             operands are pixel-major, the reduction runs over pixels; Q rows
//                             optionally shifted by a 3x3 tap = implicit im2col for conv wgrad)
//   isp_relu_mask_colsum      g = dy * (y > 0), bias gradient = column sums of g
//   isp_classifier_bwd        dx = (x > 0) * g[m] * w[c], dw, db of the 1x1 classifier
//   isp_resize_bilinear_ac_nhwc_bwd / _nchw_f32_bwd   adjoints of the align_corners bilinear resizes
//   isp_layernorm_bwd         d LayerNorm / dx with recomputed statistics, accumulating into the fp32 gradient stream
#include "isp_common.h"

namespace {

typedef __attribute__((ext_vector_type(4))) short s16x4;

// ---------------------------------------------------------------------------------------
// TN GEMM via transposed LDS reads.  Block tile 128(n) x 128(j), K-step = 64 pixels, 4 waves
// (2 x 2, 64 x 64 each = 4 x 4 MFMA 16x16x32), 2-stage LDS ring filled by LDS-DMA.  Both tiles are
// stored [pixel][channel] (256-byte rows); every MFMA fragment (8 consecutive pixels of one
// channel) is two ds_read_b64_tr_b16.  16-B chunks are XOR-swizzled with
// f(row) = ((row&3)<<1) ^ (((row>>3)&1)<<3) so the 8 rows a half-wave touches hit 8 distinct
// 32-byte windows.  Split-K over pixels; fp32 atomicAdd of the finished tile.
constexpr int TBK = 64, TBN = 128;
constexpr int T_TILE = TBK * TBN * 2;  // 16 KiB per operand tile
constexpr int T_LDS = 4 * T_TILE;

__device__ __forceinline__ int tswz(int row, int chunk) { return chunk ^ (((row & 3) << 1) | (((row >> 3) & 1) << 3)); }

struct TapShift {  // Q rows are pixels of an NHWC map shifted by a 3x3 tap (dy, dx); zero outside
    int H, W, dy, dx, enabled;
};

__global__ __launch_bounds__(256, 2) void tn_gemm_kernel(const bf16_t* __restrict__ P, long ldp,
                                                          const bf16_t* __restrict__ Q, long ldq, float* __restrict__ out,
                                                          long ldo, long M, int N, int J, long m_per_block, TapShift ts,
                                                          int tiles_n, int tiles_j) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    int bid = blockIdx.x;
    const int tj = bid % tiles_j;
    bid /= tiles_j;
    const int tn = bid % tiles_n;
    const long split = bid / tiles_n;
    const long m_begin = split * m_per_block;
    long m_end = m_begin + m_per_block;
    if (m_end > M) m_end = M;
    if (m_begin >= m_end) return;
    const int n0 = tn * TBN, j0 = tj * TBN;

    // --- DMA assignment: piece q = 4 pixel rows x 256 B; wave w takes pieces w, w+4, w+8, w+12
    const int prow = lane >> 4, pch = lane & 15;
    long mrow[4];
    int yy[4], xx[4], pcol[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = (wid + 4 * i) * 4 + prow;
        mrow[i] = m_begin + row;
        pcol[i] = tswz(row, pch) * 8;
        if (ts.enabled) {
            const unsigned hw = (unsigned)ts.H * (unsigned)ts.W;
            const unsigned rem = (unsigned)((unsigned long)mrow[i] % hw);
            yy[i] = (int)(rem / (unsigned)ts.W);
            xx[i] = (int)(rem % (unsigned)ts.W);
        }
    }
    auto stage = [&](char* buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool live = mrow[i] < m_end;
            const int nn = n0 + pcol[i], jj = j0 + pcol[i];
            const void* ps = (live && nn < N) ? (const void*)(P + (size_t)mrow[i] * ldp + nn) : (const void*)g_isp_zero16;
            glds16(ps, buf + (wid + 4 * i) * 1024);
            bool qok = live && jj < J;
            long qrow = mrow[i];
            if (ts.enabled) {
                qok = qok && (unsigned)(yy[i] + ts.dy) < (unsigned)ts.H && (unsigned)(xx[i] + ts.dx) < (unsigned)ts.W;
                qrow += (long)ts.dy * ts.W + ts.dx;
            }
            const void* qs = qok ? (const void*)(Q + (size_t)qrow * ldq + jj) : (const void*)g_isp_zero16;
            glds16(qs, buf + T_TILE + (wid + 4 * i) * 1024);
            mrow[i] += TBK;
            if (ts.enabled) {
                xx[i] += TBK;
                while (xx[i] >= ts.W) xx[i] -= ts.W, ++yy[i];
                while (yy[i] >= ts.H) yy[i] -= ts.H;
            }
        }
    };

    // --- fragment geometry: 16-lane group reads 4 pixel rows x 16 channels
    const int wn = wid >> 1, wj = wid & 1;
    const int g = lane >> 4, gi = lane & 15, gq = gi >> 2, gp = gi & 3;
    f32x4 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto frag = [&](const char* tile, int ks, int col0) {  // 8 pixels (k = 32ks + 8g + j) of channel col0 + gi
        bf16x8 r;
        s16x4 part[2];
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            const int row = ks * 32 + 8 * g + 4 * jj + gq;
            const int col = col0 + 4 * gp;
            part[jj] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                (ISP_LDS s16x4*)(tile + row * 256 + tswz(row, col >> 3) * 16 + (col & 7) * 2));
        }
        r = bf16x8{part[0][0], part[0][1], part[0][2], part[0][3], part[1][0], part[1][1], part[1][2], part[1][3]};
        return r;
    };
    auto compute = [&](const char* buf) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 fa[4], fb[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                fa[t] = frag(buf, ks, wn * 64 + t * 16);
                fb[t] = frag(buf + T_TILE, ks, wj * 64 + t * 16);
            }
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[a], fb[b], acc[a][b], 0, 0, 0);
        }
    };

    const long nk = (m_end - m_begin + TBK - 1) / TBK;
    char* buf0 = smem;
    char* buf1 = smem + 2 * T_TILE;
    stage(buf0);
    __syncthreads();
    long kt = 0;
    for (; kt + 2 <= nk; kt += 2) {
        stage(buf1);
        compute(buf0);
        __syncthreads();
        if (kt + 2 < nk) stage(buf0);
        compute(buf1);
        __syncthreads();
    }
    if (kt < nk) compute(buf0);

    // D[row = n][col = j]: lane col j = .. + (lane&15), rows n = .. + 4*(lane>>4) + i
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int j = j0 + wj * 64 + b * 16 + gi;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int n = n0 + wn * 64 + a * 16 + 4 * g + i;
                if (n < N && j < J) atomicAdd(out + (size_t)n * ldo + j, acc[a][b][i]);
            }
        }
}

// ---------------------------------------------------------------------------------------
// Row-streaming elementwise kernels with per-column reductions.  A block of 256 threads covers
// ROWS_PER_BLOCK rows x all N columns: thread -> (row slot = tid / (N/8), 8 columns = tid % (N/8)),
// so every 16-byte access is coalesced along the row; per-column partial sums are combined in LDS and
// leave the block as ONE atomicAdd per column.
constexpr int ROWS_PER_BLOCK = 256;
constexpr int RED_MAX_N = 2048;  // N/8 <= 256 threads

template <int NSUM, class F>
__device__ __forceinline__ void stream_rows_colsum(long M, int N, float* const (&sums)[NSUM], F&& body) {
    __shared__ float red[NSUM][RED_MAX_N];
    const int nc8 = N >> 3;
    const int slots = 256 / nc8;  // row slots per pass
    const int slot = threadIdx.x / nc8, c8 = threadIdx.x - slot * nc8;
    for (int i = threadIdx.x; i < NSUM * RED_MAX_N; i += 256) (&red[0][0])[i] = 0.f;
    __syncthreads();
    if (slot < slots) {
        float s[NSUM][8];
#pragma unroll
        for (int k = 0; k < NSUM; ++k)
#pragma unroll
            for (int e = 0; e < 8; ++e) s[k][e] = 0.f;
        const long r0 = (long)blockIdx.x * ROWS_PER_BLOCK;
        const long r1 = r0 + ROWS_PER_BLOCK < M ? r0 + ROWS_PER_BLOCK : M;
        for (long r = r0 + slot; r < r1; r += slots) body(r, c8, s);
#pragma unroll
        for (int k = 0; k < NSUM; ++k)
            if (sums[k])
#pragma unroll
                for (int e = 0; e < 8; ++e) atomicAdd(&red[k][c8 * 8 + e], s[k][e]);  // LDS atomics
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NSUM; ++k)
        if (sums[k])
            for (int n = threadIdx.x; n < N; n += 256) atomicAdd(sums[k] + n, red[k][n]);
}

// g = dy * (y > 0) (bf16), colsum[n] += sum_m g[m][n]
__global__ __launch_bounds__(256) void relu_mask_colsum_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ y,
                                                                bf16_t* __restrict__ g, float* __restrict__ colsum,
                                                                long M, int N) {
    float* const sums[1] = {colsum};
    stream_rows_colsum<1>(M, N, sums, [&](long r, int c8, float (&s)[1][8]) {
        const uint4 a = *reinterpret_cast<const uint4*>(dy + r * N + c8 * 8);
        const uint4 b = *reinterpret_cast<const uint4*>(y + r * N + c8 * 8);
        const unsigned* pa = &a.x;
        const unsigned* pb = &b.x;
        unsigned o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float ylo = __uint_as_float(pb[e] << 16), yhi = __uint_as_float(pb[e] & 0xffff0000u);
            const unsigned lo = ylo > 0.f ? (pa[e] & 0xffffu) : 0u, hi = yhi > 0.f ? (pa[e] & 0xffff0000u) : 0u;
            o[e] = lo | hi;
            s[0][2 * e] += __uint_as_float(lo << 16);
            s[0][2 * e + 1] += __uint_as_float(hi);
        }
        *reinterpret_cast<uint4*>(g + r * N + c8 * 8) = make_uint4(o[0], o[1], o[2], o[3]);
    });
}

// ---------------------------------------------------------------------------------------
// Train-mode BatchNorm2d on an NHWC bf16 map seen as [M = B*H*W pixels] x [C channels].  The reference's
// `self.net.train()` (core/training/trainer.py:214,431) also flips the FROZEN upsamplers' BatchNorm2d layers
// (LiFT.py:19-24,71-76,88; loftup/loftup.py:58,63) to batch statistics; this is that forward and its backward.
//   stats:  sums[0][c] += sum_m x, sums[1][c] += sum_m x^2      (fp32 partials per 256-row block, then atomics)
//   apply:  y = act((x - mean) * rstd * gamma + beta),  mean = S1/M, var = S2/M - mean^2 (biased), rstd = rsqrt(var+eps);
//           block 0 also emits the running statistics a torch BatchNorm would hold afterwards
//           (new = (1-momentum) old + momentum {mean, var * M/(M-1)}).
//   bwd:    g = dy * [y > 0];  dx = gamma * rstd * (g - mean_m(g) - xhat * mean_m(g * xhat)),  xhat = (x - mean) * rstd
//           (two passes: the two column sums, then the elementwise combine).
__device__ __forceinline__ void bn_mean_rstd(const float* __restrict__ sums, int C, int c, long M, float eps, float& mean,
                                             float& rstd) {
    const double m = (double)sums[c] / (double)M;
    const double var = fmax((double)sums[C + c] / (double)M - m * m, 0.0);
    mean = (float)m;
    rstd = (float)(1.0 / sqrt(var + (double)eps));
}

__global__ __launch_bounds__(256) void bn_stats_kernel(const bf16_t* __restrict__ x, float* __restrict__ sums, long M, int C) {
    float* const out[2] = {sums, sums + C};
    stream_rows_colsum<2>(M, C, out, [&](long r, int c8, float (&s)[2][8]) {
        const uint4 a = *reinterpret_cast<const uint4*>(x + r * C + c8 * 8);
        const unsigned* pa = &a.x;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float lo = __uint_as_float(pa[e] << 16), hi = __uint_as_float(pa[e] & 0xffff0000u);
            s[0][2 * e] += lo;
            s[0][2 * e + 1] += hi;
            s[1][2 * e] += lo * lo;
            s[1][2 * e + 1] += hi * hi;
        }
    });
}

__global__ __launch_bounds__(256) void bn_apply_kernel(const bf16_t* __restrict__ x, const float* __restrict__ sums,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        bf16_t* __restrict__ y, long M, int C, float eps, int relu,
                                                        const float* __restrict__ run_mean, const float* __restrict__ run_var,
                                                        float* __restrict__ new_mean, float* __restrict__ new_var, int c_real,
                                                        float momentum) {
    const int cv = C >> 3;
    const long total = M * cv;
    if (blockIdx.x == 0 && new_mean)
        for (int c = threadIdx.x; c < c_real; c += 256) {
            const double m = (double)sums[c] / (double)M;
            const double var = fmax((double)sums[C + c] / (double)M - m * m, 0.0);
            const double unbiased = M > 1 ? var * (double)M / (double)(M - 1) : var;
            new_mean[c] = (float)((1.0 - momentum) * run_mean[c] + momentum * m);
            new_var[c] = (float)((1.0 - momentum) * run_var[c] + momentum * unbiased);
        }
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int c8 = (int)(idx % cv);
        const uint4 a = *reinterpret_cast<const uint4*>(x + idx * 8);
        const unsigned* pa = &a.x;
        unsigned o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float v[2] = {__uint_as_float(pa[e] << 16), __uint_as_float(pa[e] & 0xffff0000u)};
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int c = c8 * 8 + 2 * e + k;
                float mean, rstd;
                bn_mean_rstd(sums, C, c, M, eps, mean, rstd);
                v[k] = (v[k] - mean) * rstd * gamma[c] + beta[c];
                if (relu) v[k] = fmaxf(v[k], 0.f);
            }
            o[e] = pack2bf(v[0], v[1]);
        }
        *reinterpret_cast<uint4*>(y + idx * 8) = make_uint4(o[0], o[1], o[2], o[3]);
    }
}

__global__ __launch_bounds__(256) void bn_bwd_stats_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x,
                                                            const bf16_t* __restrict__ y, const float* __restrict__ sums,
                                                            float* __restrict__ gsums, long M, int C, float eps) {
    float* const out[2] = {gsums, gsums + C};
    stream_rows_colsum<2>(M, C, out, [&](long r, int c8, float (&s)[2][8]) {
        const uint4 a = *reinterpret_cast<const uint4*>(dy + r * C + c8 * 8);
        const uint4 b = *reinterpret_cast<const uint4*>(x + r * C + c8 * 8);
        uint4 m = make_uint4(0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u);  // ones: no ReLU mask
        if (y) m = *reinterpret_cast<const uint4*>(y + r * C + c8 * 8);
        const unsigned *pa = &a.x, *pb = &b.x, *pm = &m.x;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float g[2] = {__uint_as_float(pm[e] << 16) > 0.f ? __uint_as_float(pa[e] << 16) : 0.f,
                                __uint_as_float(pm[e] & 0xffff0000u) > 0.f ? __uint_as_float(pa[e] & 0xffff0000u) : 0.f};
            const float xv[2] = {__uint_as_float(pb[e] << 16), __uint_as_float(pb[e] & 0xffff0000u)};
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                float mean, rstd;
                bn_mean_rstd(sums, C, c8 * 8 + 2 * e + k, M, eps, mean, rstd);
                s[0][2 * e + k] += g[k];
                s[1][2 * e + k] += g[k] * (xv[k] - mean) * rstd;
            }
        }
    });
}

__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x,
                                                            const bf16_t* __restrict__ y, const float* __restrict__ sums,
                                                            const float* __restrict__ gsums, const float* __restrict__ gamma,
                                                            bf16_t* __restrict__ dx, long M, int C, float eps) {
    const int cv = C >> 3;
    const long total = M * cv;
    const float inv_m = 1.f / (float)M;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int c8 = (int)(idx % cv);
        const uint4 a = *reinterpret_cast<const uint4*>(dy + idx * 8);
        const uint4 b = *reinterpret_cast<const uint4*>(x + idx * 8);
        uint4 m = make_uint4(0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u);
        if (y) m = *reinterpret_cast<const uint4*>(y + idx * 8);
        const unsigned *pa = &a.x, *pb = &b.x, *pm = &m.x;
        unsigned o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float g[2] = {__uint_as_float(pm[e] << 16) > 0.f ? __uint_as_float(pa[e] << 16) : 0.f,
                                __uint_as_float(pm[e] & 0xffff0000u) > 0.f ? __uint_as_float(pa[e] & 0xffff0000u) : 0.f};
            const float xv[2] = {__uint_as_float(pb[e] << 16), __uint_as_float(pb[e] & 0xffff0000u)};
            float r[2];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int c = c8 * 8 + 2 * e + k;
                float mean, rstd;
                bn_mean_rstd(sums, C, c, M, eps, mean, rstd);
                const float xhat = (xv[k] - mean) * rstd;
                r[k] = gamma[c] * rstd * (g[k] - gsums[c] * inv_m - xhat * gsums[C + c] * inv_m);
            }
            o[e] = pack2bf(r[0], r[1]);
        }
        *reinterpret_cast<uint4*>(dx + idx * 8) = make_uint4(o[0], o[1], o[2], o[3]);
    }
}

// ---------------------------------------------------------------------------------------
// 1x1 classifier backward, optionally fused with the ReLU mask of its input x (MASK: x is a post-ReLU conv output;
// otherwise x is a signed feature map -- the "linear" head, or a conv head with num_layers = 0 -- and dx is unmasked):
//   dx[m][c] = (!MASK || x[m][c] > 0) ? g[m] * w[c] : 0     dw[c] += sum_m g[m] * x[m][c]     db += sum_m g[m]
//   dxsum[c] += sum_m dx[m][c]   (optional: the bias gradient of the conv that produced x)
template <bool MASK>
__global__ __launch_bounds__(256) void classifier_bwd_kernel(const float* __restrict__ gl, const bf16_t* __restrict__ x,
                                                              const float* __restrict__ w, bf16_t* __restrict__ dx,
                                                              float* __restrict__ dw, float* __restrict__ db,
                                                              float* __restrict__ dxsum, long M, int C) {
    float* const sums[2] = {dw, dxsum};
    stream_rows_colsum<2>(M, C, sums, [&](long r, int c8, float (&s)[2][8]) {
        const float gm = gl[r];
        const uint4 b = *reinterpret_cast<const uint4*>(x + r * C + c8 * 8);
        const float4 w0 = *reinterpret_cast<const float4*>(w + c8 * 8), w1 = *reinterpret_cast<const float4*>(w + c8 * 8 + 4);
        const float wv[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
        const unsigned* pb = &b.x;
        unsigned o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float xlo = __uint_as_float(pb[e] << 16), xhi = __uint_as_float(pb[e] & 0xffff0000u);
            s[0][2 * e] += gm * xlo;
            s[0][2 * e + 1] += gm * xhi;
            o[e] = pack2bf(!MASK || xlo > 0.f ? gm * wv[2 * e] : 0.f, !MASK || xhi > 0.f ? gm * wv[2 * e + 1] : 0.f);
            s[1][2 * e] += __uint_as_float(o[e] << 16);
            s[1][2 * e + 1] += __uint_as_float(o[e] & 0xffff0000u);
        }
        *reinterpret_cast<uint4*>(dx + r * C + c8 * 8) = make_uint4(o[0], o[1], o[2], o[3]);
    });
    if (threadIdx.x < 64) {  // db: one wave sums the block's rows of gl
        const long r0 = (long)blockIdx.x * ROWS_PER_BLOCK;
        float sg = 0.f;
        for (long r = r0 + threadIdx.x; r < r0 + ROWS_PER_BLOCK && r < M; r += 64) sg += gl[r];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) sg += __shfl_xor(sg, o);
        if (threadIdx.x == 0) atomicAdd(db, sg);
    }
}

// ---------------------------------------------------------------------------------------
// Adjoint of the align_corners bilinear resize [B,h,w,C] -> [B,H,W,C]: gather form, one thread
// per (source pixel, 8 channels) sweeping the destination pixels whose 2x2 support contains it.
__global__ __launch_bounds__(256) void bilinear_bwd_kernel(const bf16_t* __restrict__ dout, bf16_t* __restrict__ din,
                                                            int h, int w, int H, int W, int C, float sy, float sx,
                                                            long total) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int cv = C >> 3, c8 = (int)(idx % cv);
    long t = idx / cv;
    const int x = (int)(t % w);
    t /= w;
    const int y = (int)(t % h);
    const long b = t / h;
    // destination rows Y with floor(sy*Y) in {y-1, y}; generous bounds, exact test inside
    const int Y0 = sy > 0.f ? max(0, (int)floorf((float)(y - 1) / sy) - 1) : 0;
    const int Y1 = sy > 0.f ? min(H - 1, (int)ceilf((float)(y + 1) / sy) + 1) : H - 1;
    const int X0 = sx > 0.f ? max(0, (int)floorf((float)(x - 1) / sx) - 1) : 0;
    const int X1 = sx > 0.f ? min(W - 1, (int)ceilf((float)(x + 1) / sx) + 1) : W - 1;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int Y = Y0; Y <= Y1; ++Y) {
        const float fy = sy * (float)Y;
        const int y0 = (int)fy, y1 = min(y0 + 1, h - 1);
        const float ly = fy - (float)y0;
        float wy = 0.f;
        if (y0 == y) wy += 1.f - ly;
        if (y1 == y) wy += ly;
        if (wy == 0.f) continue;
        for (int X = X0; X <= X1; ++X) {
            const float fx = sx * (float)X;
            const int x0 = (int)fx, x1 = min(x0 + 1, w - 1);
            const float lx = fx - (float)x0;
            float wx = 0.f;
            if (x0 == x) wx += 1.f - lx;
            if (x1 == x) wx += lx;
            if (wx == 0.f) continue;
            const uint4 u = *reinterpret_cast<const uint4*>(dout + (((size_t)b * H + Y) * W + X) * C + c8 * 8);
            const unsigned* p = &u.x;
            const float wgt = wy * wx;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                acc[2 * e] += wgt * __uint_as_float(p[e] << 16);
                acc[2 * e + 1] += wgt * __uint_as_float(p[e] & 0xffff0000u);
            }
        }
    }
    *reinterpret_cast<uint4*>(din + idx * 8) =
        make_uint4(pack2bf(acc[0], acc[1]), pack2bf(acc[2], acc[3]), pack2bf(acc[4], acc[5]), pack2bf(acc[6], acc[7]));
}

// Same adjoint for planar fp32 maps [N,h,w] <- [N,H,W] (the logits resize of iseg_base_model.py:75-80,
// which sits between the head and the loss when the upsampler output is smaller than the image).
__global__ __launch_bounds__(256) void bilinear_bwd_planar_kernel(const float* __restrict__ dout, float* __restrict__ din,
                                                                   int h, int w, int H, int W, float sy, float sx,
                                                                   long total) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int x = (int)(idx % w);
    const int y = (int)((idx / w) % h);
    const long n = idx / ((long)w * h);
    const int Y0 = sy > 0.f ? max(0, (int)floorf((float)(y - 1) / sy) - 1) : 0;
    const int Y1 = sy > 0.f ? min(H - 1, (int)ceilf((float)(y + 1) / sy) + 1) : H - 1;
    const int X0 = sx > 0.f ? max(0, (int)floorf((float)(x - 1) / sx) - 1) : 0;
    const int X1 = sx > 0.f ? min(W - 1, (int)ceilf((float)(x + 1) / sx) + 1) : W - 1;
    float acc = 0.f;
    for (int Y = Y0; Y <= Y1; ++Y) {
        const float fy = sy * (float)Y;
        const int y0 = (int)fy, y1 = min(y0 + 1, h - 1);
        const float ly = fy - (float)y0;
        float wy = 0.f;
        if (y0 == y) wy += 1.f - ly;
        if (y1 == y) wy += ly;
        if (wy == 0.f) continue;
        for (int X = X0; X <= X1; ++X) {
            const float fx = sx * (float)X;
            const int x0 = (int)fx, x1 = min(x0 + 1, w - 1);
            const float lx = fx - (float)x0;
            float wx = 0.f;
            if (x0 == x) wx += 1.f - lx;
            if (x1 == x) wx += lx;
            if (wx != 0.f) acc += wy * wx * dout[((size_t)n * H + Y) * W + X];
        }
    }
    din[idx] = acc;
}

// LayerNorm affine gradients: dgamma[c] += sum_r gy[r][c] * xhat[r][c], dbeta[c] += sum_r gy[r][c] (statistics
// recomputed).  Block = 32 rows: every wave first reduces mean / rstd of 8 rows into LDS, then the threads sweep the
// columns.  Needed only where a LayerNorm is TRAINED (the simple-ViT click encoder, simple_ViT.py:18-155).
template <typename TX>
__global__ __launch_bounds__(256) void layernorm_wgrad_kernel(const TX* __restrict__ x, long ld_x,
                                                              const bf16_t* __restrict__ gy, long ld_gy,
                                                              float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                              long rows, int D, float eps) {
    __shared__ float s_mean[32], s_rstd[32];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const long r0 = (long)blockIdx.x * 32;
    auto ld = [&](long r, int c) -> float {
        if constexpr (sizeof(TX) == 4) return x[r * ld_x + c];
        else return bf2f(x[r * ld_x + c]);
    };
    for (int k = 0; k < 8; ++k) {
        const long r = r0 + wv * 8 + k;
        float sum = 0.f, sq = 0.f;
        if (r < rows) {
            for (int c = lane; c < D; c += 64) sum += ld(r, c);
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
            const float mean = sum / (float)D;
            for (int c = lane; c < D; c += 64) {
                const float d = ld(r, c) - mean;
                sq += d * d;
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
            if (lane == 0) s_mean[wv * 8 + k] = mean, s_rstd[wv * 8 + k] = 1.0f / sqrtf(sq / (float)D + eps);
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < D; c += 256) {
        float dg = 0.f, db = 0.f;
        for (int k = 0; k < 32 && r0 + k < rows; ++k) {
            const float g = bf2f(gy[(r0 + k) * ld_gy + c]);
            dg += g * (ld(r0 + k, c) - s_mean[k]) * s_rstd[k];
            db += g;
        }
        atomicAdd(dgamma + c, dg);
        atomicAdd(dbeta + c, db);
    }
}

// ---------------------------------------------------------------------------------------
// LayerNorm backward w.r.t. the input (weights frozen), one wave per INPUT row:
//   g^ = gy * gamma,  x^ = (x - mean) * rstd,  dx = rstd * (g^ - mean(g^) - x^ * mean(g^ x^)).
// Statistics are recomputed from the fp32 row.  gx (fp32, the gradient of the residual stream) is
// accumulated (gx += dx) or overwritten; a bf16 copy of the updated row feeds the next GEMM.
// With group_out > 0 the forward dropped `skip` leading rows of each (group_out+skip)-row group
// (the cls token, DINOv2.py:533-534): those rows receive zero.
template <typename TX, int MAXV>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const TX* __restrict__ x, long ld_x,
                                                             const bf16_t* __restrict__ gy, long ld_gy,
                                                             const float* __restrict__ gamma, float* __restrict__ gx,
                                                             long ld_gx, bf16_t* __restrict__ gx16, long ld_g16, long rows,
                                                             int D, float eps, int group_out, int skip, int accumulate) {
    const int lane = threadIdx.x & 63;
    const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    long rg = r;  // row of gy
    bool has_g = true;
    if (group_out > 0) {
        const long grp = r / (group_out + skip), idx = r % (group_out + skip);
        has_g = idx >= skip;
        rg = grp * group_out + (idx - skip);
    }
    const int nchunk = D >> 2;
    const TX* xr = x + r * ld_x;
    float4 v[MAXV], g[MAXV];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = lane + i * 64;
        v[i] = g[i] = make_float4(0, 0, 0, 0);
        if (c < nchunk) {
            if constexpr (sizeof(TX) == 4) {
                v[i] = *reinterpret_cast<const float4*>(xr + c * 4);
            } else {
                const uint2 u = *reinterpret_cast<const uint2*>(xr + c * 4);
                v[i] = make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u),
                                   __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u));
            }
            sum += (v[i].x + v[i].y) + (v[i].z + v[i].w);
            if (has_g) {
                const uint2 u = *reinterpret_cast<const uint2*>(gy + rg * ld_gy + c * 4);
                const float4 gm = *reinterpret_cast<const float4*>(gamma + c * 4);
                g[i] = make_float4(__uint_as_float(u.x << 16) * gm.x, __uint_as_float(u.x & 0xffff0000u) * gm.y,
                                   __uint_as_float(u.y << 16) * gm.z, __uint_as_float(u.y & 0xffff0000u) * gm.w);
            }
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
            v[i].x -= mean, v[i].y -= mean, v[i].z -= mean, v[i].w -= mean;
            sq += (v[i].x * v[i].x + v[i].y * v[i].y) + (v[i].z * v[i].z + v[i].w * v[i].w);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
    const float rstd = 1.0f / sqrtf(sq / (float)D + eps);
    float sg = 0.f, sgx = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        v[i].x *= rstd, v[i].y *= rstd, v[i].z *= rstd, v[i].w *= rstd;  // x^
        sg += (g[i].x + g[i].y) + (g[i].z + g[i].w);
        sgx += (g[i].x * v[i].x + g[i].y * v[i].y) + (g[i].z * v[i].z + g[i].w * v[i].w);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        sg += __shfl_xor(sg, o);
        sgx += __shfl_xor(sgx, o);
    }
    const float mg = sg / (float)D, mgx = sgx / (float)D;
    // padding columns [D, ld): zero (they feed zero weight rows of the next GEMM; garbage would be NaN * 0)
    if (!accumulate)
        for (int c = nchunk + lane; c < (int)(ld_gx >> 2); c += 64)
            *reinterpret_cast<float4*>(gx + r * ld_gx + c * 4) = make_float4(0, 0, 0, 0);
    if (gx16)
        for (int c = nchunk + lane; c < (int)(ld_g16 >> 2); c += 64)
            *reinterpret_cast<uint2*>(gx16 + r * ld_g16 + c * 4) = make_uint2(0, 0);
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = lane + i * 64;
        if (c < nchunk) {
            float4 d = make_float4(rstd * (g[i].x - mg - v[i].x * mgx), rstd * (g[i].y - mg - v[i].y * mgx),
                                   rstd * (g[i].z - mg - v[i].z * mgx), rstd * (g[i].w - mg - v[i].w * mgx));
            float* gp = gx + r * ld_gx + c * 4;
            if (accumulate) {
                const float4 old = *reinterpret_cast<const float4*>(gp);
                d.x += old.x, d.y += old.y, d.z += old.z, d.w += old.w;
            }
            *reinterpret_cast<float4*>(gp) = d;
            if (gx16) *reinterpret_cast<uint2*>(gx16 + r * ld_g16 + c * 4) = make_uint2(pack2bf(d.x, d.y), pack2bf(d.z, d.w));
        }
    }
}

}  // namespace

extern "C" int isp_tn_gemm_bf16_atomic(const void* P, long ldp, const void* Q, long ldq, float* out, long ldo, long M,
                                       int N, int J, int shift_H, int shift_W, int shift_dy, int shift_dx,
                                       int splits, void* stream) {
    ISP_CHECK_ARG(P && Q && out && M > 0 && N > 0 && J > 0 && ldp >= N && ldq >= J && ldo >= J && splits > 0);
    ISP_CHECK_ARG(ldp % 8 == 0 && ldq % 8 == 0 && N % 8 == 0 && J % 8 == 0);
    TapShift ts{shift_H, shift_W, shift_dy, shift_dx, (shift_H > 0 && shift_W > 0) ? 1 : 0};
    if (ts.enabled) ISP_CHECK_ARG(M % ((long)shift_H * shift_W) == 0 && abs(shift_dy) <= 1 && abs(shift_dx) <= 1);
    const int tiles_n = (N + TBN - 1) / TBN, tiles_j = (J + TBN - 1) / TBN;
    long per = ((M + splits - 1) / splits + TBK - 1) / TBK * TBK;
    const long nsplit = (M + per - 1) / per;
    const long nwg = nsplit * tiles_n * tiles_j;
    ISP_CHECK_ARG(nwg <= 0x7fffffffL);
    static bool attr_done = false;
    if (!attr_done) {
        if (hipFuncSetAttribute((const void*)tn_gemm_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, T_LDS) != hipSuccess)
            return ISP_ERR_LAUNCH;
        attr_done = true;
    }
    tn_gemm_kernel<<<(unsigned)nwg, 256, T_LDS, (hipStream_t)stream>>>((const bf16_t*)P, ldp, (const bf16_t*)Q, ldq, out,
                                                                     ldo, M, N, J, per, ts, tiles_n, tiles_j);
    return isp_launch_status();
}

extern "C" int isp_relu_mask_colsum(const void* dy, const void* y, void* g, float* colsum, long M, int N, void* stream) {
    ISP_CHECK_ARG(dy && y && g && M > 0 && N > 0 && N % 8 == 0 && N <= RED_MAX_N);
    relu_mask_colsum_kernel<<<(unsigned)((M + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK), 256, 0, (hipStream_t)stream>>>(
        (const bf16_t*)dy, (const bf16_t*)y, (bf16_t*)g, colsum, M, N);
    return isp_launch_status();
}

extern "C" int isp_bn_train_stats(const void* x, float* sums, long M, int C, void* stream) {
    ISP_CHECK_ARG(x && sums && M > 0 && C > 0 && C % 8 == 0 && C <= RED_MAX_N);
    bn_stats_kernel<<<(unsigned)((M + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK), 256, 0, (hipStream_t)stream>>>((const bf16_t*)x, sums, M, C);
    return isp_launch_status();
}

extern "C" int isp_bn_train_apply(const void* x, const float* sums, const float* gamma, const float* beta, void* y, long M, int C,
                                  float eps, int relu, const float* running_mean, const float* running_var, float* new_mean,
                                  float* new_var, int c_real, float momentum, void* stream) {
    ISP_CHECK_ARG(x && sums && gamma && beta && y && M > 0 && C > 0 && C % 8 == 0 && C <= RED_MAX_N);
    ISP_CHECK_ARG(!new_mean || (running_mean && running_var && new_var && c_real > 0 && c_real <= C));
    const long total = M * (C / 8);
    const unsigned grid = (unsigned)((total + 255) / 256 < 65536 ? (total + 255) / 256 : 65536);
    bn_apply_kernel<<<grid, 256, 0, (hipStream_t)stream>>>((const bf16_t*)x, sums, gamma, beta, (bf16_t*)y, M, C, eps, relu,
                                                          running_mean, running_var, new_mean, new_var, c_real, momentum);
    return isp_launch_status();
}

extern "C" int isp_bn_train_bwd(const void* dy, const void* x, const void* y, const float* sums, const float* gamma,
                                float* gsums, void* dx, long M, int C, float eps, void* stream) {
    ISP_CHECK_ARG(dy && x && sums && gamma && gsums && dx && M > 0 && C > 0 && C % 8 == 0 && C <= RED_MAX_N);
    bn_bwd_stats_kernel<<<(unsigned)((M + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK), 256, 0, (hipStream_t)stream>>>(
        (const bf16_t*)dy, (const bf16_t*)x, (const bf16_t*)y, sums, gsums, M, C, eps);
    const long total = M * (C / 8);
    const unsigned grid = (unsigned)((total + 255) / 256 < 65536 ? (total + 255) / 256 : 65536);
    bn_bwd_apply_kernel<<<grid, 256, 0, (hipStream_t)stream>>>((const bf16_t*)dy, (const bf16_t*)x, (const bf16_t*)y, sums, gsums,
                                                              gamma, (bf16_t*)dx, M, C, eps);
    return isp_launch_status();
}

extern "C" int isp_classifier_bwd(const float* grad_logits, const void* x, const float* w, void* dx, float* dw,
                                  float* db, float* dx_colsum, long M, int C, int relu_mask, void* stream) {
    ISP_CHECK_ARG(grad_logits && x && w && dx && dw && db && M > 0 && C > 0 && C % 8 == 0);
    ISP_CHECK_ARG(C <= RED_MAX_N);
    const unsigned grid = (unsigned)((M + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK);
    if (relu_mask)
        classifier_bwd_kernel<true><<<grid, 256, 0, (hipStream_t)stream>>>(grad_logits, (const bf16_t*)x, w, (bf16_t*)dx, dw, db,
                                                                          dx_colsum, M, C);
    else
        classifier_bwd_kernel<false><<<grid, 256, 0, (hipStream_t)stream>>>(grad_logits, (const bf16_t*)x, w, (bf16_t*)dx, dw, db,
                                                                           dx_colsum, M, C);
    return isp_launch_status();
}

extern "C" int isp_resize_bilinear_ac_nhwc_bwd(const void* dout, void* din, int B, int h, int w, int H, int W, int C,
                                               void* stream) {
    ISP_CHECK_ARG(dout && din && B > 0 && h > 0 && w > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0);
    const float sy = H > 1 ? (float)(h - 1) / (float)(H - 1) : 0.f;
    const float sx = W > 1 ? (float)(w - 1) / (float)(W - 1) : 0.f;
    const long total = (long)B * h * w * (C / 8);
    bilinear_bwd_kernel<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(
        (const bf16_t*)dout, (bf16_t*)din, h, w, H, W, C, sy, sx, total);
    return isp_launch_status();
}

template <typename TX>
static int launch_ln_bwd(const void* x, long ld_x, const void* gy, long ld_gy, const float* gamma, float* gx, long ld_gx,
                         void* g16, long ld_g16, long rows, int D, float eps, int group_out, int skip, int accumulate,
                         hipStream_t s) {
    const int nchunk = D / 4;
    dim3 grid((unsigned)((rows + 3) / 4));
#define LNB_CASE(MV)                                                                                                       \
    layernorm_bwd_kernel<TX, MV><<<grid, 256, 0, s>>>((const TX*)x, ld_x, (const bf16_t*)gy, ld_gy, gamma, gx, ld_gx,     \
                                                      (bf16_t*)g16, ld_g16, rows, D, eps, group_out, skip, accumulate)
    if (nchunk <= 64) LNB_CASE(1);
    else if (nchunk <= 128) LNB_CASE(2);
    else if (nchunk <= 256) LNB_CASE(4);
    else return ISP_ERR_UNSUPPORTED;
#undef LNB_CASE
    return isp_launch_status();
}

extern "C" int isp_layernorm_bwd(const void* x, int x_dtype, long ld_x, const void* gy, long ld_gy, const float* gamma,
                                 float* gx, long ld_gx, void* gx_bf16, long ld_g16, long rows, int D, float eps,
                                 int group_out, int skip, int accumulate, void* stream) {
    ISP_CHECK_ARG(x && gy && gamma && gx && rows > 0 && D > 0 && D % 4 == 0 && group_out >= 0 && skip >= 0);
    ISP_CHECK_ARG(group_out == 0 || rows % (group_out + skip) == 0);
    if (ld_x <= 0) ld_x = D;
    if (ld_gy <= 0) ld_gy = D;
    if (ld_gx <= 0) ld_gx = D;
    if (ld_g16 <= 0) ld_g16 = D;
    ISP_CHECK_ARG(ld_x >= D && ld_gy >= D && ld_gx >= D && ld_g16 >= D);
    ISP_CHECK_ARG(ld_x % 4 == 0 && ld_gy % 4 == 0 && ld_gx % 4 == 0 && ld_g16 % 4 == 0);
    hipStream_t s = (hipStream_t)stream;
    if (x_dtype == ISP_F32)
        return launch_ln_bwd<float>(x, ld_x, gy, ld_gy, gamma, gx, ld_gx, gx_bf16, ld_g16, rows, D, eps, group_out, skip, accumulate, s);
    if (x_dtype == ISP_BF16)
        return launch_ln_bwd<bf16_t>(x, ld_x, gy, ld_gy, gamma, gx, ld_gx, gx_bf16, ld_g16, rows, D, eps, group_out, skip, accumulate, s);
    return ISP_ERR_UNSUPPORTED;
}

extern "C" int isp_resize_bilinear_ac_nchw_f32_bwd(const float* dout, float* din, long planes, int h, int w, int H, int W,
                                                   void* stream) {
    ISP_CHECK_ARG(dout && din && planes > 0 && h > 0 && w > 0 && H > 0 && W > 0);
    const float sy = H > 1 ? (float)(h - 1) / (float)(H - 1) : 0.f;
    const float sx = W > 1 ? (float)(w - 1) / (float)(W - 1) : 0.f;
    const long total = planes * h * w;
    bilinear_bwd_planar_kernel<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(dout, din, h, w, H, W, sy,
                                                                                               sx, total);
    return isp_launch_status();
}

extern "C" int isp_layernorm_wgrad(const void* x, int x_dtype, long ld_x, const void* gy, long ld_gy, float* dgamma,
                                   float* dbeta, long rows, int D, float eps, void* stream) {
    ISP_CHECK_ARG(x && gy && dgamma && dbeta && rows > 0 && D > 0);
    if (ld_x <= 0) ld_x = D;
    if (ld_gy <= 0) ld_gy = D;
    ISP_CHECK_ARG(ld_x >= D && ld_gy >= D);
    hipStream_t s = (hipStream_t)stream;
    const unsigned grid = (unsigned)((rows + 31) / 32);
    if (x_dtype == ISP_F32)
        layernorm_wgrad_kernel<float><<<grid, 256, 0, s>>>((const float*)x, ld_x, (const bf16_t*)gy, ld_gy, dgamma, dbeta, rows, D, eps);
    else if (x_dtype == ISP_BF16)
        layernorm_wgrad_kernel<bf16_t><<<grid, 256, 0, s>>>((const bf16_t*)x, ld_x, (const bf16_t*)gy, ld_gy, dgamma, dbeta, rows, D, eps);
    else
        return ISP_ERR_UNSUPPORTED;
    return isp_launch_status();
}
