// FeatUp joint-bilateral-upsampling (JBU) stage kernels for gfx950.
//
// The arithmetic lives in a third-party package that is absent from the reference tree
// (mhamilton723/FeatUp, reached through reference core/model/upsamplers/JBUFeatUp.py:30-32);
// it is restated from the published algorithm (JBULearnedRange / JBUStack / AdaptiveConv),
// see oracle/upsamplers.py::_jbu_stage.  One x2 stage =
//   guidance (fp32 NCHW, 3 ch) --adaptive_avg_pool--> G [B,3,GH,GW]
//   proj   = conv1x1(gelu(conv1x1(G)))                  [B,GH,GW,32]      (isp_jbu_range_proj)
//   kernel = softmax_t(temp * <proj(nbr_t), proj>) * gauss_t, renormalised,
//            += 0.1 * fixup_mlp([kernel, G])            [49] per pixel
//   out    = sum_t kernel_t * hr(reflect(p + t)),  hr = bicubic_x2(source)
//
// MI355X formulation: bicubic x2 and the 7x7 stencil are both linear in `source`, so their
// composition is ONE 8x8 stencil on the LOW-RES source per output pixel.  isp_jbu_kernels
// emits that composite kernel (bf16, [B,GH,GW,8 rows,16 circular column slots]); isp_jbu_apply
// evaluates it with MFMA as banded GEMMs over 16-pixel strips.  The x2 feature map `hr` (6.4 GB
// at 512^2 x 384 ch x 32 images) never exists, in HBM or anywhere else.
#include "isp_common.h"

namespace {
// ---- fp16 inside the stack.  The composite-kernel records, the fix-up MLP operands and the feature maps between the
// stages are IEEE half (11 significant bits) rather than bf16 (8): every stage re-rounds its input, its kernel weights and
// its output, and with bf16 those roundings (1.3e-3 / 1.3e-3 / 1.7e-3 relative each, measured per stage by
// tools/diag_jbu_precision.py) made the four stages the largest contributor to the bench workload's logit error.  All
// values here are bounded -- kernel weights sum to 1 per pixel, features are convex-ish combinations of LayerNorm-ed
// tokens -- so half's range is ample; the MFMA rate of the f16 forms is that of the bf16 ones.  The stack's input is
// converted bf16 -> f16 (exact) and its last stage writes bf16 for the head's convolution.
// (pack2h / h_lo / h_hi / pack2o: isp_common.h)

constexpr int R = 3, DIA = 7, TAPS = 49, KEY = 32;

__device__ __forceinline__ int reflect(int i, int n) {  // F.pad(mode="reflect")
    i = i < 0 ? -i : i;
    return i >= n ? 2 * (n - 1) - i : i;
}

// --------------------------------------------------------------------------------------
// F.adaptive_avg_pool2d on NCHW fp32 planes (window = [floor(i*in/out), ceil((i+1)*in/out)) ).  One block per (plane,
// output row): the row's vertical window is wave-uniform, a thread walks output columns ox, ox + blockDim, ... (neighbouring
// threads read neighbouring input columns), no 64-bit index arithmetic per element (the first version spent most of
// its 118 us per launch on four long divisions per output).
__global__ __launch_bounds__(256) void adaptive_avg_pool_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                                 int H, int W, int OH, int OW) {
    const int oy = blockIdx.x, plane = blockIdx.y;
    const int y0 = (oy * H) / OH, y1 = ((oy + 1) * H + OH - 1) / OH;
    const float* p = in + (size_t)plane * H * W;
    float* o = out + ((size_t)plane * OH + oy) * OW;
    for (int ox = threadIdx.x; ox < OW; ox += blockDim.x) {
        const int x0 = (ox * W) / OW, x1 = ((ox + 1) * W + OW - 1) / OW;
        float s = 0.f;
        for (int y = y0; y < y1; ++y)
            for (int x = x0; x < x1; ++x) s += p[(size_t)y * W + x];
        o[ox] = s / (float)((y1 - y0) * (x1 - x0));
    }
}

// --------------------------------------------------------------------------------------
// range_proj, exact form: everything in fp32 on the VALU, erf GELU -- for the fp32 checking mode (core/model/precise.py) and
// the stage-by-stage fp32 derivation (jbu_f32.hip).  One thread per pixel.
__global__ __launch_bounds__(256) void jbu_range_proj_f32_kernel(const float* __restrict__ G, float* __restrict__ proj,
                                                                  const float* __restrict__ w0, const float* __restrict__ b0,
                                                                  const float* __restrict__ w3, const float* __restrict__ b3,
                                                                  const float* __restrict__ drop, long HW, long total) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const long b = idx / HW, p = idx - b * HW;
    const float g0 = G[(b * 3 + 0) * HW + p], g1 = G[(b * 3 + 1) * HW + p], g2 = G[(b * 3 + 2) * HW + p];
    float hid[KEY];
#pragma unroll
    for (int j = 0; j < KEY; ++j)
        hid[j] = gelu_erf(fmaf(w0[j * 3 + 2], g2, fmaf(w0[j * 3 + 1], g1, fmaf(w0[j * 3 + 0], g0, b0[j])))) *
                 (drop ? drop[b * KEY + j] : 1.0f);
    float4* o = reinterpret_cast<float4*>(proj + idx * KEY);
#pragma unroll
    for (int m4 = 0; m4 < KEY / 4; ++m4) {
        float r[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int m = m4 * 4 + q;
            float s = b3[m];
#pragma unroll
            for (int j = 0; j < KEY; ++j) s = fmaf(w3[m * KEY + j], hid[j], s);
            r[q] = s;
        }
        o[m4] = make_float4(r[0], r[1], r[2], r[3]);
    }
}

// range_proj: 1x1 (3 -> 32), GELU, 1x1 (32 -> 32); output NHWC IEEE half (jbu_kernels stages it as half anyway: 64 instead
// of 128 bytes per pixel written here and read there, with its halo, 2.1 times).  Block = 256 pixels.  Layer 1 (96 FMAs + 32 GELUs) is
// computed by the pixel's own thread; layer 2 (1024 FMAs per pixel on the VALU in the first version: 0.55 ms at 512^2 x 32)
// is one f16 MFMA pair per 16 pixels: hidden rows go through LDS as half [pixel][32] (80-byte pitch: the operand reads
// of 16 consecutive pixels x one k-chunk touch every bank once), D[out][pixel] = W3 . H^T leaves as float4 per lane.
__global__ __launch_bounds__(256) void jbu_range_proj_kernel(const float* __restrict__ G, unsigned short* __restrict__ proj,
                                                              const float* __restrict__ w0, const float* __restrict__ b0,
                                                              const float* __restrict__ w3, const float* __restrict__ b3,
                                                              const float* __restrict__ drop, long HW, long total) {
    __shared__ __attribute__((aligned(16))) char s_hid[256 * 80];
    __shared__ __attribute__((aligned(16))) _Float16 s_w3[KEY * KEY];
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    for (int i = threadIdx.x; i < KEY * KEY; i += 256) s_w3[i] = (_Float16)w3[i];
    {
        const long ic = idx < total ? idx : total - 1;
        const long b = ic / HW, p = ic - b * HW;
        const float g0 = G[(b * 3 + 0) * HW + p], g1 = G[(b * 3 + 1) * HW + p], g2 = G[(b * 3 + 2) * HW + p];
#pragma unroll
        for (int c = 0; c < KEY / 8; ++c) {
            float h[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int j = c * 8 + e;
                h[e] = gelu_sig5(fmaf(w0[j * 3 + 2], g2, fmaf(w0[j * 3 + 1], g1, fmaf(w0[j * 3 + 0], g0, b0[j]))));
                if (drop) h[e] *= drop[b * KEY + j];  // train-mode Dropout2d(0.1) behind the GELU: per (image, channel) 0 or 1/0.9
            }
            *reinterpret_cast<uint4*>(s_hid + threadIdx.x * 80 + c * 16) =
                make_uint4(pack2h(h[0], h[1]), pack2h(h[2], h[3]), pack2h(h[4], h[5]), pack2h(h[6], h[7]));
        }
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, li = lane & 15, lq = lane >> 4;
    f16x8_t wf[2];  // A operand: W3[out = 16 ot + li][k = 8 lq ..]
#pragma unroll
    for (int ot = 0; ot < 2; ++ot) wf[ot] = *reinterpret_cast<const f16x8_t*>(s_w3 + (ot * 16 + li) * KEY + lq * 8);
    float4 bias[2];
#pragma unroll
    for (int ot = 0; ot < 2; ++ot) bias[ot] = *reinterpret_cast<const float4*>(b3 + ot * 16 + lq * 4);
#pragma unroll
    for (int pt = 0; pt < 4; ++pt) {  // the wave's 64 pixels in 4 tiles of 16
        const int px = wv * 64 + pt * 16 + li;
        const f16x8_t hb = *reinterpret_cast<const f16x8_t*>(s_hid + px * 80 + lq * 16);
        const long gi = (long)blockIdx.x * 256 + px;
#pragma unroll
        for (int ot = 0; ot < 2; ++ot) {
            const f32x4 d = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[ot], hb, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
            // D[out = 16 ot + 4 lq + j][pixel = px]
            if (gi < total)
                *reinterpret_cast<uint2*>(proj + gi * KEY + ot * 16 + lq * 4) =
                    make_uint2(pack2h(d[0] + bias[ot].x, d[1] + bias[ot].y), pack2h(d[2] + bias[ot].z, d[3] + bias[ot].w));
        }
    }
}

// --------------------------------------------------------------------------------------
// Per-pixel 7x7 kernels.  Block = 32x8 pixels; the 38x14 reflect-padded proj tile is staged in LDS as IEEE half in four
// k-chunk planes [k/8][pixel][8 halfs] (see the kernel body: both MFMA operands of the range logits are then conflict-free
// ds_read_b128).  The block is 32 wide so that each 16-lane group of such a read stays inside one tile row -- a 16x16
// block mixed two rows per group and 47 % of its LDS cycles were bank conflicts (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE).
typedef __attribute__((ext_vector_type(2))) float f32x2;
constexpr int TSX = 32, TSY = 8, HALOX = TSX + 2 * R, HALOY = TSY + 2 * R;
constexpr int PLANE = ((HALOY * HALOX * 16 + 255) / 256) * 256;  // one k-chunk plane of the half proj tile, 256-B multiple
constexpr int LOG_PITCH = 53;  // floats per pixel of the logit staging: odd (column reads by pixel are conflict-free) and
                              // 4 * (pitch - 1) = 16 mod 32, so the four centre groups of a band store hit every bank twice
constexpr int LOG_STAGE = 32 * LOG_PITCH * 4 + 128;             // per-wave logit staging [32 px][53] f32 + dump slots
constexpr int KERNELS_LDS_MAIN = 65536;                         // the later phases reuse the first 64 KiB (MLP / record staging)
// behind it: the spatial Gaussian (49 f32) and the block's slice of the composite-kernel tables -- bys rows of its 8 pixel
// rows, bxs rows of its 32 columns -- staged once (the per-pixel float4 table loads from L2, ~70 per pixel at 2 waves per
// SIMD, were half of the kernel's s_waitcnt time)
constexpr int TAB_GAUSS = 0, TAB_BY = 256, TAB_BX = TAB_BY + TSY * DIA * 8 * 4, KERNELS_LDS_TABLES = TAB_BX + TSX * DIA * 16 * 4;
static_assert(4 * PLANE + 4 * LOG_STAGE <= KERNELS_LDS_MAIN, "logit phase fits the later phases' LDS");

// Composite-kernel tables (host-built, depend only on the output size):
//   bys[y][ty][ry]  : weight of window row ry (src row base_y(y)+ry) in hr row reflect(y+ty-3)
//   bxs[x][tx][slot]: same for columns, indexed by the CIRCULAR slot (src col & 15)
// base(y) = ((y-4)>>1) - 1;  an hr index q reads src rows ((q-1)>>1)-1 .. +2 with the cubic
// (A=-0.75) weights at t = 0.75 (q even) / 0.25 (q odd).
// The fix-up MLP ([k(49), G(3)] -> 49 GELU -> 49) runs on MFMA: the block's 256 pixels form the
// N dimension of two 64x64 (zero-padded) GEMMs, D[unit][pixel] = W . X^T, with X / hidden / result
// staged as bf16 rows in the LDS region the proj tile no longer needs.  Wave w owns pixels
// 64w..64w+63 end to end, so only wave-local LDS ordering is involved after the one barrier.
__device__ __forceinline__ int mlp_off(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

// BLEND: the block's 32 x 8 records are not stored; the 28 x 7 records of the bilinearly resized grid that depend on
// them (and on them only, see jbu_blend_kernel below) are blended from the staged records and stored instead
// ([B, OH, OW, 9, 16]): the stage's own 256-byte records (2.1 GB at 512^2 x 32) never reach HBM.
template <bool BLEND>
__global__ __launch_bounds__(256, 2) void jbu_kernels_kernel(const void* __restrict__ proj_any, int proj_f16,
                                                           const float* __restrict__ G,
                                                           bf16_t* __restrict__ kout, const bf16_t* __restrict__ f0w,
                                                           const float* __restrict__ f0b, const bf16_t* __restrict__ f3w,
                                                           const float* __restrict__ f3b, const float* __restrict__ bys,
                                                           const float* __restrict__ bxs, float temp, float inv2s2,
                                                           int GH, int GW, int OH, int OW, float rsy, float rsx,
                                                           const float* __restrict__ drop) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int b = blockIdx.z, ty0 = blockIdx.y * TSY, tx0 = blockIdx.x * TSX;
    const long HW = (long)GH * GW;
    // ---- range logits <proj(p), proj(p + t)> for the 49 taps, on MFMA.  (The first version formed them on the VALU: 784
    // packed FMAs and 392 ds_read_b128 per pixel -- every pixel re-read its 48 neighbours' 128-byte vectors -- which made
    // the kernel LDS- and VALU-bound.)  The 38 x 14 reflect-padded proj tile is staged as IEEE half in four chunk planes
    // [k/8][pixel][8 halfs]: 16 consecutive pixels of a tile row x one k-chunk are 256 contiguous bytes, so both MFMA
    // operands (16 centres, 16 neighbours; lane = (pixel, k-chunk)) are conflict-free ds_read_b128.  Per pixel row and
    // 16-pixel strip, D[centre c][neighbour n] = one v_mfma_f32_16x16x32_f16 per (dy, neighbour group): 14 MFMAs cover the
    // 16 x 49 logits (the band n - c in [0, 6] of two 16 x 16 tiles); the band entries go through a wave-private LDS
    // staging [32 px][49] (odd pitch: conflict-free column reads) to the thread that owns the pixel.
    char* const planes = smem;                              // 4 x PLANE bytes
    char* const s_log = smem + 4 * PLANE;                   // 4 waves x [32][49] f32
    for (int i = threadIdx.x; i < HALOY * HALOX * (KEY / 8); i += 256) {
        const int c8 = i % (KEY / 8), pix = i / (KEY / 8);
        const int py = pix / HALOX, px = pix % HALOX;
        const int gy = reflect(min(ty0 + py - R, GH - 1 + R), GH), gx = reflect(min(tx0 + px - R, GW - 1 + R), GW);
        const size_t at = ((size_t)b * HW + (size_t)gy * GW + gx) * KEY + c8 * 8;
        uint4 h8;
        if (proj_f16) {  // (block-uniform) the product route: jbu_range_proj already wrote halfs
            h8 = *reinterpret_cast<const uint4*>(static_cast<const unsigned short*>(proj_any) + at);
        } else {
            const float* src = static_cast<const float*>(proj_any) + at;
            const float4 v0 = *reinterpret_cast<const float4*>(src), v1 = *reinterpret_cast<const float4*>(src + 4);
            h8 = make_uint4(pack2h(v0.x, v0.y), pack2h(v0.z, v0.w), pack2h(v1.x, v1.y), pack2h(v1.z, v1.w));
        }
        *reinterpret_cast<uint4*>(planes + c8 * PLANE + pix * 16) = h8;
    }
    // spatial Gaussian of the 49 taps (pixel-independent): computed once per block, read back as LDS broadcasts
    float* const s_gauss = reinterpret_cast<float*>(smem + KERNELS_LDS_MAIN + TAB_GAUSS);
    float* const s_by = reinterpret_cast<float*>(smem + KERNELS_LDS_MAIN + TAB_BY);  // [8 rows][7][8]
    float* const s_bx = reinterpret_cast<float*>(smem + KERNELS_LDS_MAIN + TAB_BX);  // [7][4 float4][32 cols]
    for (int i = threadIdx.x; i < TSY * DIA * 8 / 4; i += 256) {
        const int r = i / (DIA * 8 / 4);
        reinterpret_cast<float4*>(s_by)[i] =
            reinterpret_cast<const float4*>(bys + (size_t)min(ty0 + r, GH - 1) * (DIA * 8))[i - r * (DIA * 8 / 4)];
    }
    // (stored column-fastest, [float4 j of the row][32 cols]: the 32 lanes of a tile row read float4 j of 32 DIFFERENT columns --
    // row-major [col][28 float4] put lanes 448 bytes apart, 4 lanes per bank group: SQ_LDS_BANK_CONFLICT was 40 % of the LDS cycles)
    for (int i = threadIdx.x; i < TSX * DIA * 16 / 4; i += 256) {
        const int c = i & (TSX - 1), j = i / TSX;  // lane-linear LDS writes; the table rows come from L2
        reinterpret_cast<float4*>(s_bx)[i] =
            reinterpret_cast<const float4*>(bxs + (size_t)min(tx0 + c, GW - 1) * (DIA * 16))[j];
    }
    if (threadIdx.x < TAPS) {
        const int t = threadIdx.x;
        const float dy = -1.f + (float)(t / DIA) * (2.f / (DIA - 1)), dx = -1.f + (float)(t % DIA) * (2.f / (DIA - 1));
        s_gauss[t] = __expf(-(dx * dx + dy * dy) * inv2s2);
    }
    __syncthreads();
    const int lx = threadIdx.x & (TSX - 1), ly = threadIdx.x / TSX;
    const int y = min(ty0 + ly, GH - 1), x = min(tx0 + lx, GW - 1);  // out-of-image lanes compute a clamped pixel

    float k[TAPS];
    float mx = -INFINITY;
    {
        const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
        const int li = lane & 15, lq = lane >> 4;
        float* const stg = reinterpret_cast<float*>(s_log + wv * LOG_STAGE);
        const char* const pl = planes + lq * PLANE;  // the lane's k-chunk plane
#pragma unroll 1
        for (int rr = 0; rr < 2; ++rr) {  // the wave's two pixel rows
            const int row = 2 * wv + rr;
#pragma unroll
            for (int st = 0; st < 2; ++st) {  // two 16-pixel strips
                const f16x8_t ctr = *reinterpret_cast<const f16x8_t*>(pl + ((row + R) * HALOX + 16 * st + R + li) * 16);
                // all 14 products of the strip first (independent: no MFMA waits on another), then the band entries.  The
                // stores are unconditional -- entries outside the band go to a dump slot -- because 56 predicated stores per
                // strip became 56 exec-mask branches.
#pragma unroll
                for (int d0 = 0; d0 < DIA; d0 += 4) {  // tap rows in two groups (4 + 3): 8 independent MFMAs, 32 result registers
                    f32x4 d[4][2];
#pragma unroll
                    for (int dd = 0; dd < 4; ++dd)
#pragma unroll
                        for (int q = 0; q < 2; ++q) {
                            if (d0 + dd >= DIA) continue;
                            // neighbour n = 16q + li of this strip = halo column 16 st + n (columns past the tile: clamped,
                            // their products fall outside the band)
                            const int ncol = min(16 * st + 16 * q + li, HALOX - 1);
                            const f16x8_t nb = *reinterpret_cast<const f16x8_t*>(pl + ((row + d0 + dd) * HALOX + ncol) * 16);
                            d[dd][q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ctr, nb, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                        }
                    // lane: neighbour n = 16q + li, centres c = 4 lq + e  ->  tap tx = n - c
#pragma unroll
                    for (int q = 0; q < 2; ++q)
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int c = 4 * lq + e, tx = 16 * q + li - c;
                            const bool in_band = tx >= 0 && tx < DIA;
                            const int base = in_band ? (16 * st + c) * LOG_PITCH + tx : 32 * LOG_PITCH + (lane & 31);
#pragma unroll
                            for (int dd = 0; dd < 4; ++dd)
                                if (d0 + dd < DIA) stg[base + (in_band ? (d0 + dd) * DIA : 0)] = d[dd][q][e];
                        }
                }
            }
            // (written and read by the same wave: ordered by the compiler's lgkmcnt wait)
            if ((lane >> 5) == rr) {
                const float* mine = stg + (lane & 31) * LOG_PITCH;
#pragma unroll
                for (int t = 0; t < TAPS; ++t) {
                    k[t] = mine[t] * temp;
                    mx = fmaxf(mx, k[t]);
                }
            }
        }
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
        k[t] = k[t] * inv * s_gauss[t];
        sum2 += k[t];
    }
    const float inv2 = 1.f / fmaxf(sum2, 1e-7f);
#pragma unroll
    for (int t = 0; t < TAPS; ++t) k[t] *= inv2;
    // fixup MLP on [k(49), G(3)]: 52 -> 49 (GELU) -> 49, added with weight 0.1 -- on MFMA
    const long p = (long)y * GW + x;
    __syncthreads();  // every wave is done with the proj tile: its LDS is reused below
    char* s_x = smem;            // [256 px][64] bf16: X, later the result F
    char* s_h = smem + 32768;    // [256 px][64] bf16: hidden
    {
        const float g0 = G[((size_t)b * 3 + 0) * HW + p], g1 = G[((size_t)b * 3 + 1) * HW + p],
                    g2 = G[((size_t)b * 3 + 2) * HW + p];
        const int row = threadIdx.x;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            float e[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int i = c * 8 + q;
                e[q] = i < TAPS ? k[i < TAPS ? i : 0] : (i == TAPS ? g0 : (i == TAPS + 1 ? g1 : (i == TAPS + 2 ? g2 : 0.f)));
            }
            *reinterpret_cast<uint4*>(s_x + mlp_off(row, c)) =
                make_uint4(pack2h(e[0], e[1]), pack2h(e[2], e[3]), pack2h(e[4], e[5]), pack2h(e[6], e[7]));
        }
    }
    {
        const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
        const int fr = lane & 15, fq = lane >> 4;
#pragma unroll 1
        for (int layer = 0; layer < 2; ++layer) {
            const bf16_t* wsrc = layer == 0 ? f0w : f3w;
            const float* bsrc = layer == 0 ? f0b : f3b;
            const char* src = layer == 0 ? s_x : s_h;
            char* dst = layer == 0 ? s_h : s_x;
            f16x8_t wf[4][2];  // A operand: W[unit = 16*ot + fr][k = 32*ks + 8*fq + j]
#pragma unroll
            for (int ot = 0; ot < 4; ++ot)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
                    wf[ot][ks] = *reinterpret_cast<const f16x8_t*>(wsrc + (ot * 16 + fr) * 64 + ks * 32 + fq * 8);
            // (the layer's biases once, not per (pixel tile, unit tile): inside the loops each was a global load with its own
            //  wait -- 32 L2 round trips in series per workgroup)
            float4 bias4[4];
#pragma unroll
            for (int ot = 0; ot < 4; ++ot) bias4[ot] = *reinterpret_cast<const float4*>(bsrc + ot * 16 + fq * 4);
#pragma unroll
            for (int pt = 0; pt < 4; ++pt) {
                const int row = wv * 64 + pt * 16 + fr;  // B operand: X[pixel = row][k..k+7]
                const f16x8_t x0 = *reinterpret_cast<const f16x8_t*>(src + mlp_off(row, fq));
                const f16x8_t x1 = *reinterpret_cast<const f16x8_t*>(src + mlp_off(row, 4 + fq));
#pragma unroll
                for (int ot = 0; ot < 4; ++ot) {
                    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[ot][0], x0, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[ot][1], x1, acc, 0, 0, 0);
                    // D[unit = 16*ot + 4*fq + j][pixel = row]
                    const float4 bb = bias4[ot];
                    float r0 = acc[0] + bb.x, r1 = acc[1] + bb.y, r2 = acc[2] + bb.z, r3 = acc[3] + bb.w;
                    if (layer == 0) {
                        r0 = gelu_sig5(r0), r1 = gelu_sig5(r1), r2 = gelu_sig5(r2), r3 = gelu_sig5(r3);
                        if (drop) {  // train-mode Dropout2d(0.1) of the fix-up MLP's hidden units: [B][64] multipliers
                            const float4 dm = *reinterpret_cast<const float4*>(drop + (size_t)b * 64 + ot * 16 + fq * 4);
                            r0 *= dm.x, r1 *= dm.y, r2 *= dm.z, r3 *= dm.w;
                        }
                    }
                    *reinterpret_cast<uint2*>(dst + mlp_off(row, ot * 2 + (fq >> 1)) + (fq & 1) * 8) =
                        make_uint2(pack2h(r0, r1), pack2h(r2, r3));
                }
            }
        }
    }
    {
        const int row = threadIdx.x;
#pragma unroll
        for (int c = 0; c < 7; ++c) {
            const uint4 u = *reinterpret_cast<const uint4*>(s_x + mlp_off(row, c));
            const unsigned* q = &u.x;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int i = c * 8 + 2 * e;
                if (i < TAPS) k[i] += 0.1f * h_lo(q[e]);
                if (i + 1 < TAPS) k[i + 1] += 0.1f * h_hi(q[e]);
            }
        }
    }
    // composite 8x8 kernel on the source grid: rows first (hrow[ry][tx] = sum_ty by[ty][ry] k[ty][tx],
    // k is dead afterwards), then the two halves of the 16 circular column slots
    const float* byp = s_by + ly * (DIA * 8);    // = bys[y], bxs[x] (clamped like y, x), from the block's LDS copy
    const float* bxp = s_bx + lx * 4;  // float4 j of this column at bxp + j * (TSX * 4)
    // The 256-byte record of a pixel is staged in LDS and leaves as 16 bytes per lane with 16 lanes per record: a
    // thread storing its own record directly issues 64 separate 16-byte segments per instruction (partial lines).
    // Wave-private staging [64 px][16 chunks], chunk slot XORed with the pixel index (conflict-free both ways).
    __syncthreads();  // every wave has read its MLP result rows: the LDS below is free
    char* const stg = smem + (threadIdx.x >> 6) * (64 * 256);
    const int sl = threadIdx.x & 63;
    // packed fp32 FMAs (v_pk_fma_f32): window rows in pairs for the row pass, circular slots in pairs for the column
    // pass; every output element sees the same fmaf sequence as the scalar form
    f32x2 hrow[4][DIA];
#pragma unroll
    for (int r2 = 0; r2 < 4; ++r2)
#pragma unroll
        for (int tx = 0; tx < DIA; ++tx) hrow[r2][tx] = f32x2{0.f, 0.f};
#pragma unroll
    for (int ty = 0; ty < DIA; ++ty) {
        const float4 c0 = *reinterpret_cast<const float4*>(byp + ty * 8);
        const float4 c1 = *reinterpret_cast<const float4*>(byp + ty * 8 + 4);
        const f32x2 by[4] = {f32x2{c0.x, c0.y}, f32x2{c0.z, c0.w}, f32x2{c1.x, c1.y}, f32x2{c1.z, c1.w}};
#pragma unroll
        for (int r2 = 0; r2 < 4; ++r2)
#pragma unroll
            for (int tx = 0; tx < DIA; ++tx) {
                const float kv = k[ty * DIA + tx];
                hrow[r2][tx] = __builtin_elementwise_fma(by[r2], f32x2{kv, kv}, hrow[r2][tx]);
            }
    }
#pragma unroll 1
    for (int half = 0; half < 2; ++half) {
        f32x2 bx[DIA][4];
#pragma unroll
        for (int tx = 0; tx < DIA; ++tx) {
            const float4 b0 = *reinterpret_cast<const float4*>(bxp + (tx * 4 + half * 2) * (TSX * 4));
            const float4 b1 = *reinterpret_cast<const float4*>(bxp + (tx * 4 + half * 2 + 1) * (TSX * 4));
            bx[tx][0] = f32x2{b0.x, b0.y}, bx[tx][1] = f32x2{b0.z, b0.w};
            bx[tx][2] = f32x2{b1.x, b1.y}, bx[tx][3] = f32x2{b1.z, b1.w};
        }
#pragma unroll
        for (int ry = 0; ry < 8; ++ry) {
            f32x2 r[4] = {f32x2{0.f, 0.f}, f32x2{0.f, 0.f}, f32x2{0.f, 0.f}, f32x2{0.f, 0.f}};
#pragma unroll
            for (int tx = 0; tx < DIA; ++tx) {
                const float hv = hrow[ry >> 1][tx][ry & 1];
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) r[s4] = __builtin_elementwise_fma(f32x2{hv, hv}, bx[tx][s4], r[s4]);
            }
            *reinterpret_cast<uint4*>(stg + sl * 256 + (((ry * 2 + half) ^ (sl & 15)) << 4)) =
                make_uint4(pack2h(r[0].x, r[0].y), pack2h(r[1].x, r[1].y), pack2h(r[2].x, r[2].y), pack2h(r[3].x, r[3].y));
        }
    }
    if constexpr (!BLEND) {
        // (written and read by the same wave: ordered by the compiler's lgkmcnt wait)  The wave's 64 pixels are two
        // runs of 32 consecutive pixels (tile rows ly = 2w, 2w+1).
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int id = j * 64 + sl, spx = id >> 4, ck = id & 15;
            const uint4 q = *reinterpret_cast<const uint4*>(stg + spx * 256 + ((ck ^ (spx & 15)) << 4));
            const int gy = ty0 + (threadIdx.x >> 6) * 2 + (spx >> 5), gx = tx0 + (spx & 31);
            if (gy < GH && gx < GW)
                *reinterpret_cast<uint4*>(kout + ((size_t)b * HW + (size_t)gy * GW + gx) * 128 + ck * 8) = q;
        }
    } else {
        __syncthreads();  // all 256 records staged
        // record of tile pixel (ly, lx), chunk c (= window row * 2 + half): wave ly/2, slot (ly & 1) * 32 + lx
        auto rec = [&](int rly, int rlx, int c) {
            const int s2 = (rly & 1) * 32 + rlx;
            return *reinterpret_cast<const uint4*>(smem + (rly >> 1) * (64 * 256) + s2 * 256 + ((c ^ (s2 & 15)) << 4));
        };
        const int oy0 = blockIdx.y * 7, ox0 = blockIdx.x * 28;
        for (int i = threadIdx.x; i < 7 * 28 * 18; i += 256) {  // item = 8 slots of one window row of one output pixel
            const int half = i & 1, r = (i >> 1) % 9, op = i / 18;
            const int Y = oy0 + op / 28, X = ox0 + op % 28;
            if (Y >= OH || X >= OW) continue;
            const float fy = rsy * (float)Y, fx = rsx * (float)X;  // bilinear_nhwc_kernel's arithmetic
            const int y0 = (int)fy, x0 = (int)fx;
            // (y0 + 1, x0 + 1 leave the tile only where they also leave the image: 8 rows <-> 7 rows)
            const int y1 = min(min(y0 + 1, GH - 1), ty0 + TSY - 1), x1 = min(min(x0 + 1, GW - 1), tx0 + TSX - 1);
            const float wy1 = fy - (float)y0, wx1 = fx - (float)x0;
            const int by0 = ((y0 - 4) >> 1) - 1;
            float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int dy = 0; dy < 2; ++dy) {
                const int yy = dy ? y1 : y0;
                const float wy = dy ? wy1 : 1.f - wy1;
                const int ry = r - ((((yy - 4) >> 1) - 1) - by0);
                if (ry < 0 || ry > 7) continue;
#pragma unroll
                for (int dx = 0; dx < 2; ++dx) {
                    const int xx = dx ? x1 : x0;
                    const float wgt = wy * (dx ? wx1 : 1.f - wx1);
                    const uint4 v = rec(yy - ty0, xx - tx0, ry * 2 + half);
                    const unsigned* q = &v.x;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        acc[2 * e] = fmaf(wgt, h_lo(q[e]), acc[2 * e]);
                        acc[2 * e + 1] = fmaf(wgt, h_hi(q[e]), acc[2 * e + 1]);
                    }
                }
            }
            *reinterpret_cast<uint4*>(kout + (((size_t)b * OH + Y) * OW + X) * 144 + r * 16 + half * 8) =
                make_uint4(pack2h(acc[0], acc[1]), pack2h(acc[2], acc[3]), pack2h(acc[4], acc[5]), pack2h(acc[6], acc[7]));
        }
    }
}

// --------------------------------------------------------------------------------------
// Apply the composite kernels: out[b,y,x,:] = sum_{ry,rx} kc[b,y,x][ry][rx] * src[b, clamp(base_y+ry),
// clamp(base_x+rx), :].  Block = 8 rows x 32 cols of output pixels (4 waves x 2 rows x 2 column strips of 16),
// all channels (looped in chunks of 64).  A 16-pixel strip's windows [base_x(x), base_x(x)+8) lie inside the 16
// source columns o' .. o'+15, o' = x0/2 - 4, and the record's 16 circular slots (src col & 15, zero outside the
// window -- the format's precondition) are exactly those 16 columns: per window row ry the strip's kernels ARE
// the [16 src cols x 16 px] B operand of v_mfma_f32_16x16x16_bf16, one aligned 8-byte load per lane straight
// from the kernel tensor, no masking.  The A operand is the transposed source row ([16 ch x 16 cols]), one
// ds_read_b64_tr_b16 from the [pixel][channel] LDS tile.  D[ch][px] lands with 4 consecutive channels per lane.
// The two column strips share one 11 x 24-pixel source tile -- the same tile the earlier 8 x 16-pixel block (K = 32
// MFMAs, 24 of 32 columns used) loaded for half the outputs: source traffic through L2 13 -> 6.6 GB per 512^2 launch
// (the kernel sits on L2 bandwidth: 22 GB moved for a 10 GB HBM floor).
//   LDS: source tile 11 rows x 24 cols x 64 ch bf16 (33 KiB, 16-B chunks XOR-swizzled for the transposed reads)
//        + wave-private output staging 4 x 64 px x 144 B (36 KiB): the wave's pixels leave as 16 bytes per lane,
//        8 lanes per 128-byte pixel segment, instead of 64 separate 8-byte segments per store.
//   The band fragments (2 rows x 2 strips x 8 window rows x 2 VGPRs) live in registers across the channel loop.
constexpr int ATH = 8, ATW = 32, ACC = 64;
constexpr int SROWS = 11, SCOLS = 24, SPIX = SROWS * SCOLS;                   // 264 source pixels
constexpr int SRC_TILE_BYTES = ((SPIX + 7) / 8) * 8 * ACC * 2;                // padded to whole 1 KiB pieces
constexpr int STG_PITCH = ACC * 2 + 16;                                        // output staging: bytes per pixel
constexpr int APPLY_LDS = SRC_TILE_BYTES + 4 * 64 * STG_PITCH;

typedef __attribute__((ext_vector_type(4))) short s16x4_t;

// 16-B chunk swizzle of the source tile by the pixel's COLUMN in the tile: a transposed read touches 16 consecutive
// columns x 32 B, which (column parity = bank half, (col >> 1) & 3 = which 32-byte quarter) spread over all banks twice
// -- the minimum for 512 bytes -- and a fragment address becomes lane constant + row * pitch (an immediate).
__device__ __forceinline__ int src_swz(int col, int chunk) { return chunk ^ (((col >> 1) & 3) << 1); }
constexpr int SROWB = 24 * 128;  // bytes per source-tile row

template <bool OUT_BF16>
__global__ __launch_bounds__(256, 2) void jbu_apply_kernel(const bf16_t* __restrict__ src, const bf16_t* __restrict__ kc,
                                                           bf16_t* __restrict__ out, int h, int w, int C, int tiles_x,
                                                           int tiles_y, int nwg) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* s_src = smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int GH = 2 * h, GW = 2 * w;
    int wg = xcd_remap(blockIdx.x, nwg);
    const int tx = wg % tiles_x;
    wg /= tiles_x;
    const int ty = wg % tiles_y, b = wg / tiles_y;
    const int y0 = ty * ATH, x0 = tx * ATW;
    const int tile_y0 = ((y0 - 4) >> 1) - 1;  // base_y(y0)
    const int tile_x0 = (x0 >> 1) - 4;        // o' of the first strip = base_x(x0) - 1 (a multiple of 4)

    // ---- band fragments: lane (px = lane&15, g = lane>>4) holds, per (row si, strip cs, window row ry), the 4 slots
    // of src cols o'(cs) + 4g .. +3, i.e. the 8-byte chunk ((o' >> 2) + g) & 3 of the record's 32-byte row ry
    const int px = lane & 15, g = lane >> 4;
    f16x4_t band[2][2][8];
#pragma unroll
    for (int si = 0; si < 2; ++si)
#pragma unroll
        for (int cs = 0; cs < 2; ++cs) {
            const int gy = min(y0 + wid * 2 + si, GH - 1), gx = min(x0 + cs * 16 + px, GW - 1);
            const int chunk = ((tile_x0 >> 2) + 2 * cs + g) & 3;
            const bf16_t* kp = kc + (((size_t)b * GH + gy) * GW + gx) * 128 + chunk * 4;
#pragma unroll
            for (int ry = 0; ry < 8; ++ry) band[si][cs][ry] = *reinterpret_cast<const f16x4_t*>(kp + ry * 16);
        }
    // ---- per-lane transposed-read geometry: a 16-lane group reads 4 src cols x 16 channels; fb = byte offset of the
    // lane's fragment piece inside a tile row, per (strip, 16-channel block)
    const int gi = lane & 15, gq = gi >> 2, gp = gi & 3;
    int fb[2][4];
#pragma unroll
    for (int cs = 0; cs < 2; ++cs)
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) {
            const int col = cs * 8 + 4 * g + gq;
            fb[cs][cb] = col * 128 + src_swz(col, cb * 2 + (gp >> 1)) * 16 + (gp & 1) * 8;
        }
    const size_t src_img = (size_t)b * h * w * C;
    // the wave's two rows (2k, 2k+1) share base_y, hence the transposed source fragments
    const int r0 = (((min(y0 + wid * 2, GH - 1) - 4) >> 1) - 1) - tile_y0;  // first window row inside the tile
    char* const stg = smem + SRC_TILE_BYTES + wid * (64 * STG_PITCH);

    for (int c0 = 0; c0 < C; c0 += ACC) {
        __syncthreads();  // previous chunk's reads done before the tile is overwritten
        // stage the source tile: 1 KiB pieces of 8 pixels x 128 B by LDS-DMA, swizzle on the source
        for (int piece = wid; piece < (SPIX + 7) / 8; piece += 4) {
            const int pix = piece * 8 + (lane >> 3);
            const int pr = pix / SCOLS, pc = pix - pr * SCOLS;
            const int sy = min(max(tile_y0 + pr, 0), h - 1), sx = min(max(tile_x0 + pc, 0), w - 1);
            const int chunk = src_swz(pc, lane & 7);
            glds16(src + src_img + ((size_t)sy * w + sx) * C + c0 + chunk * 8, s_src + piece * 1024);
        }
        __syncthreads();  // (emits vmcnt(0): the DMA has landed)
        f32x4 acc[2][2][4];
#pragma unroll
        for (int si = 0; si < 2; ++si)
#pragma unroll
            for (int cs = 0; cs < 2; ++cs)
#pragma unroll
                for (int cb = 0; cb < 4; ++cb) acc[si][cs][cb] = f32x4{0.f, 0.f, 0.f, 0.f};
        const char* const rbase = s_src + r0 * SROWB;
#pragma unroll
        for (int ry = 0; ry < 8; ++ry) {
#pragma unroll
            for (int cs = 0; cs < 2; ++cs) {
#pragma unroll
                for (int cb = 0; cb < 4; ++cb) {
                    const s16x4_t a = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (ISP_LDS s16x4_t*)(rbase + fb[cs][cb] + ry * SROWB));
                    const f16x4_t ah = __builtin_bit_cast(f16x4_t, a);
                    acc[0][cs][cb] = __builtin_amdgcn_mfma_f32_16x16x16f16(ah, band[0][cs][ry], acc[0][cs][cb], 0, 0, 0);
                    acc[1][cs][cb] = __builtin_amdgcn_mfma_f32_16x16x16f16(ah, band[1][cs][ry], acc[1][cs][cb], 0, 0, 0);
                }
            }
        }
        // D[ch = 4*(lane>>4)+j][px = lane&15] -> wave-private staging [si][cs][px] x 64 ch
#pragma unroll
        for (int si = 0; si < 2; ++si)
#pragma unroll
            for (int cs = 0; cs < 2; ++cs)
#pragma unroll
                for (int cb = 0; cb < 4; ++cb)
                    *reinterpret_cast<uint2*>(stg + ((si * 2 + cs) * 16 + px) * STG_PITCH + cb * 32 + g * 8) = make_uint2(
                        pack2o<OUT_BF16>(acc[si][cs][cb][0], acc[si][cs][cb][1]), pack2o<OUT_BF16>(acc[si][cs][cb][2], acc[si][cs][cb][3]));
        // (written and read by the same wave: ordered by the compiler's lgkmcnt wait)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int id = j * 64 + lane, sp = id >> 3, ck = id & 7;  // sp = (si*2 + cs)*16 + px: 32 pixels of row si
            const uint4 q = *reinterpret_cast<const uint4*>(stg + sp * STG_PITCH + ck * 16);
            const int oy = y0 + wid * 2 + (sp >> 5), ox = x0 + (sp & 31);
            if (oy < GH && ox < GW) *reinterpret_cast<uint4*>(out + (((size_t)b * GH + oy) * GW + ox) * C + c0 + ck * 8) = q;
        }
    }
}

// --------------------------------------------------------------------------------------
// The last JBU stage followed by the model's bilinear resize to the image size (iseg_probe_model.py:120-129), as ONE
// linear operator on the low-res source.  FeatUp's x16 output is 16/14 of the image size (512 vs 448): an output
// pixel blends the 2x2 stage pixels (y0|y1, x0|x1) with the bilinear weights, so its composite kernel is the same
// blend of their records -- the circular column slots are absolute source columns, so columns just add; the rows
// of y1 sit one window row lower when base_y(y1) = base_y(y0) + 1, which makes the blended window 9 rows.
// Because 8 stage rows map onto exactly 7 output rows (src = dst * (8m-1)/(7m-1) never leaves the 8-row group, nor
// the 16-column group of a 14-pixel strip), the apply below runs on the very same source tiles as jbu_apply_kernel
// and writes the resized map directly: the stage's 512^2 map (6.4 GB at batch 32) and the 2.3 ms resize pass disappear.

// blended records [B, OH, OW, 9, 16] bf16 from stage records [B, GH, GW, 8, 16]; a thread owns 8 slots of one row
__global__ __launch_bounds__(256) void jbu_blend_kernel(const bf16_t* __restrict__ kc, bf16_t* __restrict__ kout, int GH, int GW,
                                                        int OH, int OW, float sy, float sx, long total) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int half = (int)(idx & 1), r = (int)((idx >> 1) % 9);
    long pix = idx / 18;
    const int X = (int)(pix % OW);
    pix /= OW;
    const int Y = (int)(pix % OH);
    const int b = (int)(pix / OH);
    const float fy = sy * (float)Y, fx = sx * (float)X;  // the resize kernel's arithmetic (bilinear_nhwc_kernel)
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = min(y0 + 1, GH - 1), x1 = min(x0 + 1, GW - 1);
    const float ly = fy - (float)y0, lx = fx - (float)x0;
    const int by0 = ((y0 - 4) >> 1) - 1;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int dy = 0; dy < 2; ++dy) {
        const int yy = dy ? y1 : y0;
        const float wy = dy ? ly : 1.f - ly;
        const int ry = r - ((((yy - 4) >> 1) - 1) - by0);
        if (ry < 0 || ry > 7) continue;
#pragma unroll
        for (int dx = 0; dx < 2; ++dx) {
            const int xx = dx ? x1 : x0;
            const float wgt = wy * (dx ? lx : 1.f - lx);
            const uint4 v = *reinterpret_cast<const uint4*>(kc + (((size_t)b * GH + yy) * GW + xx) * 128 + ry * 16 + half * 8);
            const unsigned* q = &v.x;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                acc[2 * e] = fmaf(wgt, h_lo(q[e]), acc[2 * e]);
                acc[2 * e + 1] = fmaf(wgt, h_hi(q[e]), acc[2 * e + 1]);
            }
        }
    }
    *reinterpret_cast<uint4*>(kout + (((size_t)b * OH + Y) * OW + X) * 144 + r * 16 + half * 8) =
        make_uint4(pack2h(acc[0], acc[1]), pack2h(acc[2], acc[3]), pack2h(acc[4], acc[5]), pack2h(acc[6], acc[7]));
}

// Block = 7 rows x 28 cols of OUTPUT pixels (the 8 x 32 stage tile of jbu_apply_kernel), 4 waves x 2 rows x 2 strips
// of 14 pixels (MFMA columns 14, 15 idle), 9 window rows.  Same source tile, same operand forms as above.
constexpr int RTH = 7, RTW = 28, RSTRIP = 14;
// 12 staged source rows: window row 8 of the tile's last output rows lies one row past the 11 the stage needs (it only
// ever meets zero weights, but must hold finite values)
constexpr int RSROWS = 12, RSPIX = RSROWS * SCOLS, RSRC_BYTES = ((RSPIX + 7) / 8) * 8 * ACC * 2;
constexpr int RAPPLY_LDS = RSRC_BYTES + 4 * 64 * STG_PITCH;

template <bool OUT_BF16>
__global__ __launch_bounds__(256, 2) void jbu_apply_resized_kernel(const bf16_t* __restrict__ src, const bf16_t* __restrict__ kc9,
                                                                   bf16_t* __restrict__ out, int h, int w, int OH, int OW,
                                                                   int C, float sy, int tiles_x, int tiles_y, int nwg) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* s_src = smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    int wg = xcd_remap(blockIdx.x, nwg);
    const int tx = wg % tiles_x;
    wg /= tiles_x;
    const int ty = wg % tiles_y, b = wg / tiles_y;
    const int tile_y0 = ((ty * 8 - 4) >> 1) - 1;  // base_y of the stage tile's first row
    const int tile_x0 = tx * 16 - 4;              // o' of the first strip

    const int px = lane & 15, g = lane >> 4;
    f16x4_t band[2][2][9];
    int r0[2];
#pragma unroll
    for (int si = 0; si < 2; ++si) {
        const int orow = wid * 2 + si, oy = min(ty * RTH + orow, OH - 1);
        const int sy0 = (int)(sy * (float)oy);  // stage row y0 of this output row (jbu_blend_kernel's arithmetic)
        r0[si] = ((((sy0 - 4) >> 1) - 1)) - tile_y0;
#pragma unroll
        for (int cs = 0; cs < 2; ++cs) {
            const int ox = min(tx * RTW + cs * RSTRIP + px, OW - 1);
            const bool live = orow < RTH && px < RSTRIP;
            const int chunk = ((tile_x0 >> 2) + 2 * cs + g) & 3;
            const bf16_t* kp = kc9 + (((size_t)b * OH + oy) * OW + ox) * 144 + chunk * 4;
#pragma unroll
            for (int r = 0; r < 9; ++r) {
                f16x4_t v = *reinterpret_cast<const f16x4_t*>(kp + r * 16);
                if (!live) v = f16x4_t{0, 0, 0, 0};
                band[si][cs][r] = v;
            }
        }
    }
    const int gi = lane & 15, gq = gi >> 2, gp = gi & 3;
    int fb[2][4];
#pragma unroll
    for (int cs = 0; cs < 2; ++cs)
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) {
            const int col = cs * 8 + 4 * g + gq;
            fb[cs][cb] = col * 128 + src_swz(col, cb * 2 + (gp >> 1)) * 16 + (gp & 1) * 8;
        }
    const size_t src_img = (size_t)b * h * w * C;
    char* const stg = smem + RSRC_BYTES + wid * (64 * STG_PITCH);

    for (int c0 = 0; c0 < C; c0 += ACC) {
        __syncthreads();
        for (int piece = wid; piece < (RSPIX + 7) / 8; piece += 4) {
            const int pix = piece * 8 + (lane >> 3);
            const int pr = pix / SCOLS, pc = pix - pr * SCOLS;
            const int syy = min(max(tile_y0 + pr, 0), h - 1), sxx = min(max(tile_x0 + pc, 0), w - 1);
            const int chunk = src_swz(pc, lane & 7);
            glds16(src + src_img + ((size_t)syy * w + sxx) * C + c0 + chunk * 8, s_src + piece * 1024);
        }
        __syncthreads();
        f32x4 acc[2][2][4];
#pragma unroll
        for (int si = 0; si < 2; ++si)
#pragma unroll
            for (int cs = 0; cs < 2; ++cs)
#pragma unroll
                for (int cb = 0; cb < 4; ++cb) acc[si][cs][cb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 9; ++r) {
#pragma unroll
            for (int si = 0; si < 2; ++si) {
                const char* const rbase = s_src + r0[si] * SROWB;
#pragma unroll
                for (int cs = 0; cs < 2; ++cs) {
#pragma unroll
                    for (int cb = 0; cb < 4; ++cb) {
                        const s16x4_t a = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                            (ISP_LDS s16x4_t*)(rbase + fb[cs][cb] + r * SROWB));
                        acc[si][cs][cb] = __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(f16x4_t, a), band[si][cs][r],
                                                                                acc[si][cs][cb], 0, 0, 0);
                    }
                }
            }
        }
#pragma unroll
        for (int si = 0; si < 2; ++si)
#pragma unroll
            for (int cs = 0; cs < 2; ++cs)
#pragma unroll
                for (int cb = 0; cb < 4; ++cb)
                    *reinterpret_cast<uint2*>(stg + ((si * 2 + cs) * 16 + px) * STG_PITCH + cb * 32 + g * 8) = make_uint2(
                        pack2o<OUT_BF16>(acc[si][cs][cb][0], acc[si][cs][cb][1]), pack2o<OUT_BF16>(acc[si][cs][cb][2], acc[si][cs][cb][3]));
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int id = j * 64 + lane, sp = id >> 3, ck = id & 7;  // sp = (si*2 + cs)*16 + px
            const uint4 q = *reinterpret_cast<const uint4*>(stg + sp * STG_PITCH + ck * 16);
            const int orow = wid * 2 + (sp >> 5), spx = sp & 15;
            const int oy = ty * RTH + orow, ox = tx * RTW + ((sp >> 4) & 1) * RSTRIP + spx;
            if (orow < RTH && spx < RSTRIP && oy < OH && ox < OW)
                *reinterpret_cast<uint4*>(out + (((size_t)b * OH + oy) * OW + ox) * C + c0 + ck * 8) = q;
        }
    }
}

// bf16 -> f16 (exact for |v| in half's range; LayerNorm-ed ViT tokens are): the stack's input conversion
__global__ __launch_bounds__(256) void bf16_to_f16_kernel(const bf16_t* __restrict__ in, unsigned short* __restrict__ out, long n8) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n8) return;
    const uint4 u = reinterpret_cast<const uint4*>(in)[i];
    const unsigned* q = &u.x;
    unsigned o[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = pack2h(__uint_as_float(q[e] << 16), __uint_as_float(q[e] & 0xffff0000u));
    reinterpret_cast<uint4*>(out)[i] = make_uint4(o[0], o[1], o[2], o[3]);
}

}  // namespace

extern "C" int isp_bf16_to_f16(const void* in_bf16, void* out_f16, long n, void* stream) {
    ISP_CHECK_ARG(in_bf16 && out_f16 && n > 0 && n % 8 == 0);
    bf16_to_f16_kernel<<<(unsigned)((n / 8 + 255) / 256), 256, 0, (hipStream_t)stream>>>((const bf16_t*)in_bf16,
                                                                                       (unsigned short*)out_f16, n / 8);
    return isp_launch_status();
}

extern "C" int isp_adaptive_avg_pool_nchw_f32(const float* in, float* out, long planes, int H, int W, int OH, int OW,
                                              void* stream) {
    ISP_CHECK_ARG(in && out && planes > 0 && H > 0 && W > 0 && OH > 0 && OW > 0);
    ISP_CHECK_ARG(planes <= 65535 && (long)H * OH < (1L << 31) && (long)W * OW < (1L << 31));
    const unsigned threads = OW >= 256 ? 256u : (unsigned)((OW + 63) / 64 * 64);
    adaptive_avg_pool_kernel<<<dim3((unsigned)OH, (unsigned)planes), threads, 0, (hipStream_t)stream>>>(in, out, H, W, OH, OW);
    return isp_launch_status();
}

extern "C" int isp_jbu_range_proj(const float* guidance, void* proj, const float* w0, const float* b0,
                                  const float* w3, const float* b3, int B, int GH, int GW, int exact_f32,
                                  const float* drop_hidden, void* stream) {
    ISP_CHECK_ARG(guidance && proj && w0 && b0 && w3 && b3 && B > 0 && GH > 0 && GW > 0);
    const long HW = (long)GH * GW, total = HW * B;
    if (exact_f32)
        jbu_range_proj_f32_kernel<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(guidance, (float*)proj, w0,
                                                                                                    b0, w3, b3, drop_hidden, HW, total);
    else
        jbu_range_proj_kernel<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(guidance, (unsigned short*)proj,
                                                                                                w0, b0, w3, b3, drop_hidden, HW, total);
    return isp_launch_status();
}

static int launch_jbu_kernels(const void* proj, int proj_f16, const float* guidance, void* kc_bf16, const void* fix0_w, const float* fix0_b,
                              const void* fix3_w, const float* fix3_b, const float* bys, const float* bxs, float range_temp,
                              float sigma_spatial, int B, int GH, int GW, int OH, int OW, const float* drop, void* stream) {
    ISP_CHECK_ARG(proj && guidance && kc_bf16 && fix0_w && fix0_b && fix3_w && fix3_b && bys && bxs);
    ISP_CHECK_ARG(B > 0 && GH >= 4 && GW >= 4 && GH % 2 == 0 && GW % 2 == 0 && B <= 65535 && sigma_spatial != 0.f);
    const float temp = fminf(fmaxf(expf(range_temp), 1e-4f), 1e4f);
    const float inv2s2 = 1.0f / (2.f * sigma_spatial * sigma_spatial);
    const int lds = KERNELS_LDS_MAIN + KERNELS_LDS_TABLES;  // 64 KiB (MLP / record staging; the half proj tile + logit staging
                                                              // fit inside) + Gaussian + table slices = 80.2 KiB: 2 blocks per CU
    static bool attr_done = false;
    if (!attr_done) {
        if (hipFuncSetAttribute((const void*)jbu_kernels_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) !=
                hipSuccess ||
            hipFuncSetAttribute((const void*)jbu_kernels_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) !=
                hipSuccess)
            return ISP_ERR_LAUNCH;
        attr_done = true;
    }
    dim3 grid((GW + TSX - 1) / TSX, (GH + TSY - 1) / TSY, B);
    if (OH > 0) {
        const float rsy = (float)(GH - 1) / (float)(OH - 1), rsx = (float)(GW - 1) / (float)(OW - 1);
        jbu_kernels_kernel<true><<<grid, 256, lds, (hipStream_t)stream>>>(proj, proj_f16, guidance, (bf16_t*)kc_bf16, (const bf16_t*)fix0_w,
                                                                          fix0_b, (const bf16_t*)fix3_w, fix3_b, bys, bxs, temp,
                                                                          inv2s2, GH, GW, OH, OW, rsy, rsx, drop);
    } else {
        jbu_kernels_kernel<false><<<grid, 256, lds, (hipStream_t)stream>>>(proj, proj_f16, guidance, (bf16_t*)kc_bf16, (const bf16_t*)fix0_w,
                                                                           fix0_b, (const bf16_t*)fix3_w, fix3_b, bys, bxs, temp,
                                                                           inv2s2, GH, GW, 0, 0, 0.f, 0.f, drop);
    }
    return isp_launch_status();
}

extern "C" int isp_jbu_kernels(const void* proj, int proj_f16, const float* guidance, void* kc_bf16, const void* fix0_w,
                               const float* fix0_b, const void* fix3_w, const float* fix3_b, const float* bys,
                               const float* bxs, float range_temp, float sigma_spatial, int B, int GH, int GW,
                               const float* drop_hidden, void* stream) {
    return launch_jbu_kernels(proj, proj_f16, guidance, kc_bf16, fix0_w, fix0_b, fix3_w, fix3_b, bys, bxs, range_temp, sigma_spatial, B,
                              GH, GW, 0, 0, drop_hidden, stream);
}

extern "C" int isp_jbu_kernels_resized(const void* proj, int proj_f16, const float* guidance, void* kc9_bf16, const void* fix0_w,
                                       const float* fix0_b, const void* fix3_w, const float* fix3_b, const float* bys,
                                       const float* bxs, float range_temp, float sigma_spatial, int B, int GH, int GW, int OH,
                                       int OW, const float* drop_hidden, void* stream) {
    ISP_CHECK_ARG(GH >= 8 && GW >= 8 && OH > 1 && OW > 1 && GH % 8 == 0 && GW % 8 == 0);
    ISP_CHECK_ARG((long)OH * 8 == (long)GH * 7 && (long)OW * 8 == (long)GW * 7);
    return launch_jbu_kernels(proj, proj_f16, guidance, kc9_bf16, fix0_w, fix0_b, fix3_w, fix3_b, bys, bxs, range_temp, sigma_spatial, B,
                              GH, GW, OH, OW, drop_hidden, stream);
}

extern "C" int isp_jbu_blend(const void* kc_bf16, void* kc9_bf16, int B, int GH, int GW, int OH, int OW, void* stream) {
    ISP_CHECK_ARG(kc_bf16 && kc9_bf16 && B > 0 && GH >= 8 && GW >= 8 && OH > 1 && OW > 1);
    // 8 stage rows / 16 stage columns per 7 / 14 output pixels: the blended window must stay inside the stage tile
    ISP_CHECK_ARG(GH % 8 == 0 && GW % 8 == 0 && (long)OH * 8 == (long)GH * 7 && (long)OW * 8 == (long)GW * 7);
    const long total = (long)B * OH * OW * 18;
    const float sy = (float)(GH - 1) / (float)(OH - 1), sx = (float)(GW - 1) / (float)(OW - 1);
    jbu_blend_kernel<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>((const bf16_t*)kc_bf16, (bf16_t*)kc9_bf16,
                                                                                      GH, GW, OH, OW, sy, sx, total);
    return isp_launch_status();
}

extern "C" int isp_jbu_apply_resized(const void* src_nhwc_f16, const void* kc9_f16, void* out_nhwc, int B, int h, int w,
                                     int OH, int OW, int C, int out_bf16, void* stream) {
    const void *src_nhwc_bf16 = src_nhwc_f16, *kc9_bf16 = kc9_f16;
    void* out_nhwc_bf16 = out_nhwc;
    ISP_CHECK_ARG(src_nhwc_bf16 && kc9_bf16 && out_nhwc_bf16 && B > 0 && h >= 4 && w >= 4 && C > 0 && C % ACC == 0);
    ISP_CHECK_ARG((2 * h) % 8 == 0 && (2 * w) % 8 == 0 && (long)OH * 8 == (long)(2 * h) * 7 && (long)OW * 8 == (long)(2 * w) * 7);
    const int tiles_x = (OW + RTW - 1) / RTW, tiles_y = (OH + RTH - 1) / RTH;
    const long nwg = (long)tiles_x * tiles_y * B;
    ISP_CHECK_ARG(nwg <= 0x7fffffffL);
    static bool attr_done = false;
    if (!attr_done) {
        if (hipFuncSetAttribute((const void*)jbu_apply_resized_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                RAPPLY_LDS) != hipSuccess ||
            hipFuncSetAttribute((const void*)jbu_apply_resized_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                RAPPLY_LDS) != hipSuccess)
            return ISP_ERR_LAUNCH;
        attr_done = true;
    }
    const float sy = (float)(2 * h - 1) / (float)(OH - 1);
    if (out_bf16)
        jbu_apply_resized_kernel<true><<<(unsigned)nwg, 256, RAPPLY_LDS, (hipStream_t)stream>>>(
            (const bf16_t*)src_nhwc_bf16, (const bf16_t*)kc9_bf16, (bf16_t*)out_nhwc_bf16, h, w, OH, OW, C, sy, tiles_x, tiles_y,
            (int)nwg);
    else
        jbu_apply_resized_kernel<false><<<(unsigned)nwg, 256, RAPPLY_LDS, (hipStream_t)stream>>>(
            (const bf16_t*)src_nhwc_bf16, (const bf16_t*)kc9_bf16, (bf16_t*)out_nhwc_bf16, h, w, OH, OW, C, sy, tiles_x, tiles_y,
            (int)nwg);
    return isp_launch_status();
}

// --------------------------------------------------------------------------------------
// Adjoint of jbu_apply w.r.t. the source (the kernels kc depend on the guidance only):
//   gsrc[b,sy,sx,:] = sum over output pixels (y,x) and taps (ry,rx) with clamp(base_y(y)+ry) == sy and
//                     clamp(base_x(x)+rx) == sx of kc[b,y,x][ry][(base_x(x)+rx) & 15] * gout[b,y,x,:].
// Only rows y in [2sy-8, 2sy+7] (cols alike) can reach (sy,sx); away from the border each contributes one tap, at the
// border the clamped taps pile up on the edge pixel.  Gather form, one wave per source pixel, a lane owns 8 channels,
// the tap weight is wave-uniform.  Training only (models/sbd/dinov2/patch-embed_jbu.py), not tuned.
namespace {
__global__ __launch_bounds__(256) void jbu_apply_bwd_kernel(const bf16_t* __restrict__ gout, const bf16_t* __restrict__ kc,
                                                            bf16_t* __restrict__ gsrc, int h, int w, int C) {
    // Two phases per source pixel (= wave).  (1) The tap weight of each of the <= 16 x 16 output pixels that reach it, four pixels
    // per lane, into a wave-private LDS table (zero where the pixel's window misses; at the border the clamped taps add up).
    // (2) The channel loop: weight by LDS broadcast, gout rows as 16-byte loads, eight pixels per batch.  (The first version
    // looked every weight up inside the channel loop -- a dependent scalar load and four branches per pixel: 4.3 ms per launch
    // at the 512^2 stage, a fifth of the FeatUp-JBU training step.)
    __shared__ float s_wgt[4][256];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int sx = blockIdx.x * 4 + wv, sy = blockIdx.y, b = blockIdx.z;
    if (sx >= w) return;  // (wave-uniform; no block-wide barrier below)
    const int GH = 2 * h, GW = 2 * w;
    const int y_lo = 2 * sy - 8, x_lo = 2 * sx - 8;  // the 16 x 16 candidate window (may stick out of the image)
    float* const wt = s_wgt[wv];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int idx = q * 64 + lane, y = y_lo + (idx >> 4), x = x_lo + (idx & 15);
        float wgt = 0.f;
        if (y >= 0 && y < GH && x >= 0 && x < GW) {
            const int by = ((y - 4) >> 1) - 1, bx = ((x - 4) >> 1) - 1;
            int ry0 = sy - by, ry1 = sy - by, rx0 = sx - bx, rx1 = sx - bx;  // taps with clamp(base + r) == (sy, sx)
            if (sy == 0) ry0 = 0;
            if (sy == h - 1) ry1 = 7;
            if (sx == 0) rx0 = 0;
            if (sx == w - 1) rx1 = 7;
            ry0 = max(ry0, 0), ry1 = min(ry1, 7), rx0 = max(rx0, 0), rx1 = min(rx1, 7);
            const _Float16* kp = reinterpret_cast<const _Float16*>(kc + (((size_t)b * GH + y) * GW + x) * 128);
            for (int ry = ry0; ry <= ry1; ++ry)
                for (int rx = rx0; rx <= rx1; ++rx) wgt += (float)kp[ry * 16 + ((bx + rx) & 15)];
        }
        wt[idx] = wgt;
    }
    // (written and read by the same wave: the compiler's lgkmcnt wait orders them)
    const int yc0 = max(y_lo, 0), yc1 = min(y_lo + 15, GH - 1), xc0 = max(x_lo, 0), xc1 = min(x_lo + 15, GW - 1);
    for (int c0 = lane * 8; c0 < C; c0 += 512) {
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int y = yc0; y <= yc1; ++y) {
            const bf16_t* grow = gout + (((size_t)b * GH + y) * GW) * C + c0;
            const float* wrow = wt + (y - y_lo) * 16 - x_lo;
            for (int xb = xc0; xb <= xc1; xb += 8) {
                uint4 g[8];
                float wg[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int x = min(xb + u, xc1);
                    g[u] = *reinterpret_cast<const uint4*>(grow + (size_t)x * C);
                    wg[u] = xb + u <= xc1 ? wrow[x] : 0.f;
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const unsigned* q = &g[u].x;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        acc[2 * e] += wg[u] * __uint_as_float(q[e] << 16);
                        acc[2 * e + 1] += wg[u] * __uint_as_float(q[e] & 0xffff0000u);
                    }
                }
            }
        }
        *reinterpret_cast<uint4*>(gsrc + (((size_t)b * h + sy) * w + sx) * C + c0) =
            make_uint4(pack2bf(acc[0], acc[1]), pack2bf(acc[2], acc[3]), pack2bf(acc[4], acc[5]), pack2bf(acc[6], acc[7]));
    }
}
}  // namespace

extern "C" int isp_jbu_apply_bwd(const void* gout_nhwc_bf16, const void* kc_bf16, void* gsrc_nhwc_bf16, int B, int h, int w,
                                 int C, void* stream) {
    ISP_CHECK_ARG(gout_nhwc_bf16 && kc_bf16 && gsrc_nhwc_bf16 && B > 0 && h >= 2 && w >= 2 && C > 0 && C % 8 == 0);
    ISP_CHECK_ARG(B <= 65535 && h <= 65535);
    jbu_apply_bwd_kernel<<<dim3((w + 3) / 4, h, B), 256, 0, (hipStream_t)stream>>>(
        (const bf16_t*)gout_nhwc_bf16, (const bf16_t*)kc_bf16, (bf16_t*)gsrc_nhwc_bf16, h, w, C);
    return isp_launch_status();
}

extern "C" int isp_jbu_apply(const void* src_nhwc_f16, const void* kc_f16, void* out_nhwc, int B, int h, int w, int C,
                             int out_bf16, void* stream) {
    const void *src_nhwc_bf16 = src_nhwc_f16, *kc_bf16 = kc_f16;
    void* out_nhwc_bf16 = out_nhwc;
    ISP_CHECK_ARG(src_nhwc_bf16 && kc_bf16 && out_nhwc_bf16 && B > 0 && h >= 2 && w >= 2 && C > 0 && C % ACC == 0);
    const int tiles_x = (2 * w + ATW - 1) / ATW, tiles_y = (2 * h + ATH - 1) / ATH;
    const long nwg = (long)tiles_x * tiles_y * B;
    ISP_CHECK_ARG(nwg <= 0x7fffffffL);
    static bool attr_done = false;
    if (!attr_done) {
        if (hipFuncSetAttribute((const void*)jbu_apply_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, APPLY_LDS) !=
                hipSuccess ||
            hipFuncSetAttribute((const void*)jbu_apply_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, APPLY_LDS) !=
                hipSuccess)
            return ISP_ERR_LAUNCH;
        attr_done = true;
    }
    if (out_bf16)
        jbu_apply_kernel<true><<<(unsigned)nwg, 256, APPLY_LDS, (hipStream_t)stream>>>(
            (const bf16_t*)src_nhwc_bf16, (const bf16_t*)kc_bf16, (bf16_t*)out_nhwc_bf16, h, w, C, tiles_x, tiles_y, (int)nwg);
    else
        jbu_apply_kernel<false><<<(unsigned)nwg, 256, APPLY_LDS, (hipStream_t)stream>>>(
            (const bf16_t*)src_nhwc_bf16, (const bf16_t*)kc_bf16, (bf16_t*)out_nhwc_bf16, h, w, C, tiles_x, tiles_y, (int)nwg);
    return isp_launch_status();
}
