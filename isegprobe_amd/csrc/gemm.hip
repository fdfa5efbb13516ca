// bf16 MFMA GEMM core for gfx950 and its two users: dense layers (ViT linear layers,
// patch-embed, 1x1 convs) and the implicit-GEMM 3x3 convolution (seg head, LoftUp / LiFT
// convs).  One tile engine, two A-operand loaders, pluggable fused epilogues.
//
//   C[m][n] = sum_k A[m][k] * Wt[n][k]        A: activations (bf16), Wt: nn.Linear /
//                                              flattened conv weight layout [N][K] (bf16)
//
// Tile engine (v1, "2-phase" structure of cdna_hip_programming.md T3+T4 minimum form):
//   * block tile 128(M) x 128(N) x 64(K), 256 threads = 4 waves as 2(M) x 2(N),
//     each wave 64x64 = 4x4 MFMA 16x16x32 bf16 tiles, fp32 accumulate (64 acc VGPRs);
//   * A and W tiles go HBM/L2 -> LDS with 16-byte LDS-DMA (global_load_lds_dwordx4), two
//     LDS buffers, one vmcnt(0)+barrier per K-step, next tile's DMA issued before the MFMAs;
//   * LDS image: 128-byte rows (64 bf16), 16-B chunk index XOR ((row>>1)&7) -> every
//     ds_read_b128 fragment read is bank-conflict free; the DMA destination is lane-linear,
//     so the swizzle is applied to the per-lane SOURCE address (rule 21);
//   * MFMA operand roles are swapped (A-operand = weight rows, B-operand = activation rows)
//     so that a lane ends up with 4 consecutive output channels of one row: 8-byte bf16 /
//     16-byte fp32 stores and 16-byte bias loads in the epilogue;
//   * XCD-aware bijective block remap, n-tile fastest, so blocks that share an activation
//     tile run on one XCD's L2.
#include "isp_common.h"

namespace {

constexpr int BK = 64;  // K per step (one 128-byte LDS row of bf16)

// Tile configuration: block tile BM x BN, WM x WN waves; each wave owns (BM/WM) x (BN/WN) as
// TM x TN MFMA 16x16 tiles.  LDS = 2 stages x (BM + BN) x 128 B.
template <int BM_, int BN_, int WM_, int WN_, int MINW_, int NST_ = 2>
struct TileCfg {
    static constexpr int BM = BM_, BN = BN_, WM = WM_, WN = WN_, MINW = MINW_, NST = NST_;  // NST: LDS ring depth
    static constexpr int NW = WM * WN, THREADS = 64 * NW;
    static constexpr int TM = BM / WM / 16, TN = BN / WN / 16;
    static constexpr int PA = BM / 8 / NW, PW = BN / 8 / NW;  // 1 KiB DMA pieces per wave per stage
    static constexpr int A_BYTES = BM * BK * 2, W_BYTES = BN * BK * 2, STAGE = A_BYTES + W_BYTES, LDS = NST * STAGE;
    static_assert(BM % (16 * WM) == 0 && BN % (16 * WN) == 0 && (BM / 8) % NW == 0 && (BN / 8) % NW == 0, "tile");
    static_assert(NST == 2 || NST == 3, "ring depth");
};
using Cfg128 = TileCfg<128, 128, 2, 2, 2>;    // 4 waves, 64 KiB LDS, 2 blocks / CU   (short-K dense layers)
using Cfg64 = TileCfg<64, 128, 1, 4, 2>;      // 4 waves, 48 KiB LDS: twice the blocks for problems that cannot fill 256 CUs
// large-K implicit conv: 8 waves, 1 block / CU.  Measured on the seg-head conv (B=8, 448^2, C=N=384):
// 128x128 1050, 256x128 1050, 128x384 1127, 256x192 1161 TFLOP/s.
#ifndef ISP_CONV192
#define ISP_CONV192 TileCfg<256, 192, 4, 2, 2, 2>
#endif
using CfgConv192 = ISP_CONV192;                    // N % 192 == 0
using CfgConv128 = TileCfg<256, 128, 4, 2, 2, 2>;  // 96 KiB LDS
// Full-row tiles for the M = 1.6 M, N = 384 / 448 GEMMs of LoftUp's half-precision stream (8 waves as 2 x 4, 1 block / CU):
// on 256 x 192 tiles N = 448 takes three column tiles (192, 192, 64), i.e. the activation rows are staged three times
// and the third pass wastes two thirds of its MFMAs; here they are staged once per output row tile.
using CfgWide448 = TileCfg<128, 448, 2, 4, 2, 2>;  // 144 KiB LDS
using CfgWide384 = TileCfg<128, 384, 2, 4, 2, 2>;  // 128 KiB LDS
using CfgWide512 = TileCfg<128, 512, 2, 4, 2, 2>;  // 160 KiB LDS: all of a CU's (the query projection, 4 heads x 128)

// ------------------------------------------------------------------------------ A loaders
// Contract: init(slot i, global row m, 16-B source chunk) once per lane per DMA row slot;
// ptr(i) = current per-lane source address (8 bf16); advance() steps every slot one K-step on.
template <int NS>
struct DenseA {
    const bf16_t* A;
    long lda;
    long M;
    const bf16_t* cur[NS];
    template <int BM>
    __device__ __forceinline__ void init(int i, long tm, int row, int chunk) {
        const long m = tm * BM + row;
        cur[i] = A + (size_t)(m < M ? m : M - 1) * lda + chunk * 8;
    }
    template <int BM>
    __device__ __forceinline__ long out_row(long tm, int row) const {  // global output row or -1
        const long m = tm * BM + row;
        return m < M ? m : -1;
    }
    template <int BM>
    __device__ __forceinline__ bool tile_full(long tm) const { return (tm + 1) * BM <= M; }
    template <int BM>
    long num_tiles() const { return (M + BM - 1) / BM; }
    __device__ __forceinline__ const void* ptr(int i) const { return cur[i]; }
    __device__ __forceinline__ long advance() {  // returns the step of the weight-column cursor (elements)
#pragma unroll
        for (int i = 0; i < NS; ++i) cur[i] += BK;
        return BK;
    }
};

// Implicit GEMM for a 3x3, stride 1, pad 1 convolution on an NHWC bf16 map with C % 64 == 0:
// weight column k = tap*C + c, tap = (dy+1)*3 + (dx+1).  An M-tile is a 2-D patch of (BM/16) x 16 output
// pixels, not BM consecutive pixels of a row: its 9 taps touch an (BM/16+2) x 18 input patch
// (1.27x the outputs for BM=256) instead of 3 x (BM+2) (3.0x), which is what the L2 has to hold
// and what spills to the fabric.  Out-of-image taps read a 16-byte zero constant.
template <int NS>
struct Conv3x3A {
    const bf16_t* in;
    int H, W, C;
    long M;
    int cblocks;  // C / 64
    int tap, cb;  // wave-uniform iteration state
    const bf16_t* pix[NS];
    const bf16_t* cur[NS];
    int yy[NS], xx[NS];
    // K order is CHANNEL-BLOCK major, tap minor: nine consecutive K-steps sweep the 3x3 taps of one
    // 64-channel slice, so they re-touch the same 128-byte lines of the block's 18x18 input patch while
    // those are still in L2 (tap-major order walks all 384 channels per tap: 8 MB of live patches per XCD
    // against a 4 MiB L2).  Weight columns follow: k = tap*C + cb*64.
    __device__ __forceinline__ void set_tap(int i) {
        const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
        const bool ok = (unsigned)(yy[i] + dy) < (unsigned)H && (unsigned)(xx[i] + dx) < (unsigned)W;
        cur[i] = ok ? pix[i] + ((long)dy * W + dx) * C + cb * BK : reinterpret_cast<const bf16_t*>(g_isp_zero16);
    }
    int tiles_x, tiles_y;  // patches per image
    template <int BM>
    __device__ __forceinline__ void locate(long tm, int row, int& b, int& y, int& x) const {
        const unsigned per_img = (unsigned)tiles_x * (unsigned)tiles_y;
        b = (int)((unsigned long)tm / per_img);
        const unsigned t = (unsigned)((unsigned long)tm - (unsigned long)b * per_img);
        y = (int)(t / (unsigned)tiles_x) * (BM / 16) + (row >> 4);
        x = (int)(t % (unsigned)tiles_x) * 16 + (row & 15);
    }
    template <int BM>
    __device__ __forceinline__ void init(int i, long tm, int row, int chunk) {
        int b, y, x;
        locate<BM>(tm, row, b, y, x);
        y = y < H ? y : H - 1;  // rows / columns past the image edge compute a clamped pixel, never stored
        x = x < W ? x : W - 1;
        yy[i] = y;
        xx[i] = x;
        pix[i] = in + (((size_t)b * H + y) * W + x) * C + chunk * 8;
        tap = 0;
        cb = 0;
        set_tap(i);
    }
    template <int BM>
    __device__ __forceinline__ long out_row(long tm, int row) const {
        int b, y, x;
        locate<BM>(tm, row, b, y, x);
        return (y < H && x < W) ? ((long)b * H + y) * W + x : -1;
    }
    template <int BM>
    __device__ __forceinline__ bool tile_full(long tm) const {  // last row of the tile inside the image <=> all rows
        int b, y, x;
        locate<BM>(tm, BM - 1, b, y, x);
        return y < H && x < W;
    }
    template <int BM>
    long num_tiles() const { return (long)(M / ((long)H * W)) * tiles_x * tiles_y; }
    __device__ __forceinline__ const void* ptr(int i) const { return cur[i]; }
    __device__ __forceinline__ long advance() {
        long wstep = C;  // next tap, same channel block
        if (++tap == 9) {  // uniform branch
            tap = 0;
            ++cb;
            wstep = BK - 8L * C;
        }
#pragma unroll
        for (int i = 0; i < NS; ++i) set_tap(i);
        return wstep;
    }
};

// ------------------------------------------------------------------------------ epilogues
// Epilogue contract (driven by run_epilogue below), for output row m < M and columns n..n+3 < N:
//   Cols cols(n)            per-column constants (bias, LayerScale, classifier weights), loaded ONCE per wave column
//                           group before the row loop -- not once per 4x4 accumulator fragment, where every store
//                           would sit behind its own L2 round trip;
//   Pre  pre(m, n)          optional per-element operand (residual stream, saved pre-activation), loaded for a whole
//                           fragment row before any of that row's stores (a store to the same array would otherwise
//                           serialise the following load behind it);
//   unsigned row_begin(m)   optional per-row context;
//   operator()(m, n, v[4], cols [, pre] [, ctx])   /   float term(n, v[4], cols) for row-reducing epilogues.
enum { ACT_NONE = 0, ACT_RELU = 1, ACT_GELU = 2, ACT_QGELU = 3 };  // QGELU: x*sigmoid(1.702x), CLIP's QuickGELU

struct ColsBias {
    float4 b;
};
__device__ __forceinline__ ColsBias load_bias(const float* bias, int n) {  // bias may be null
    return ColsBias{bias ? *reinterpret_cast<const float4*>(bias + n) : make_float4(0, 0, 0, 0)};
}

#ifndef ISP_GELU_SIG_F16
#define ISP_GELU_SIG_F16 1
#endif
template <int ACT, bool F16 = false>  // F16: the 16-bit output is IEEE half (isp_conv3x3_nhwc_f16) instead of bf16
struct EpBiasActBf16 {  // out[m][n] = bf16(act(v + bias[n]))
    bf16_t* out;
    const float* bias;  // may be null
    long ldo;
    using Cols = ColsBias;
    __device__ __forceinline__ Cols cols(int n) const { return load_bias(bias, n); }
    // the four bf16 outputs of columns n..n+3 (kernels that stage their output tile through LDS store them themselves)
    __device__ __forceinline__ uint2 pack(long, int, const float* v, const Cols& c) const {
        float r[4] = {v[0] + c.b.x, v[1] + c.b.y, v[2] + c.b.z, v[3] + c.b.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (ACT == ACT_RELU) r[j] = fmaxf(r[j], 0.f);
            // half stream (inference): the sigmoid form of the erf GELU, max |error| 2.5e-5 = 1/20 of a half ulp at 1.0, one
            // v_exp + one v_rcp + 6 plain vector ops instead of 14 + 2 (the ViT's fc1 epilogue is 64 values per lane and tile)
            if (ACT == ACT_GELU) r[j] = (F16 && ISP_GELU_SIG_F16) ? gelu_sig5(r[j]) : gelu_erf(r[j]);
            if (ACT == ACT_QGELU) r[j] = r[j] / (1.0f + __expf(-1.702f * r[j]));
            if (F16) r[j] = __builtin_amdgcn_fmed3f(r[j], -65504.f, 65504.f);  // saturate instead of overflowing to inf
        }
        return make_uint2(pack2o<!F16>(r[0], r[1]), pack2o<!F16>(r[2], r[3]));
    }
    __device__ __forceinline__ void operator()(long m, int n, const float* v, const Cols& c) const {
        *reinterpret_cast<uint2*>(out + (size_t)m * ldo + n) = pack(m, n, v, c);
    }
};

// EpBiasActBf16 that also emits, per output row (pixel) and wave column group, the sum / sum of squares of the values it
// stores -- partial[slot][M][2], slot = n-tile * (n-waves per tile) + n-wave; column groups past N write zeros.  LoftUp's
// second convolution hands these to the first cross-attention's LN-folded query projection (EpLnFold), so the LayerNorm pass
// over the [pixels, 448] map in front of it disappears (loftup/layers.py:186-195).
template <int ACT, bool F16 = false>
struct EpBiasActStats : EpBiasActBf16<ACT, F16> {
    static constexpr bool kRowStats = true;
    float* stats;  // [slots][M][2]
    long M;
    __device__ __forceinline__ void put_stats(long m, int slot, float s1, float s2) const {
        *reinterpret_cast<float2*>(stats + ((size_t)slot * M + m) * 2) = make_float2(s1, s2);
    }
    static __device__ __forceinline__ void add(const uint2& q, float& s1, float& s2) {  // statistics of the values AS STORED
        const float q0 = F16 ? h_lo(q.x) : __uint_as_float(q.x << 16), q1 = F16 ? h_hi(q.x) : __uint_as_float(q.x & 0xffff0000u);
        const float q2 = F16 ? h_lo(q.y) : __uint_as_float(q.y << 16), q3 = F16 ? h_hi(q.y) : __uint_as_float(q.y & 0xffff0000u);
        s1 += (q0 + q1) + (q2 + q3);
        s2 += (q0 * q0 + q1 * q1) + (q2 * q2 + q3 * q3);
    }
};

__device__ __forceinline__ float act_fwd(int act, float x) {
    return act == ACT_QGELU ? x / (1.0f + __expf(-1.702f * x)) : gelu_erf(x);
}
__device__ __forceinline__ float act_grad(int act, float x) {
    if (act == ACT_QGELU) {  // d/dx x sigmoid(1.702 x) = s + 1.702 x s (1 - s)
        const float sg = 1.0f / (1.0f + __expf(-1.702f * x));
        return sg + 1.702f * x * sg * (1.0f - sg);
    }
    const float cdf = 0.5f * (1.0f + fast_erf(x * 0.70710678118654752440f));  // gelu'(x) = Phi(x) + x phi(x)
    return cdf + x * 0.39894228040143267794f * __expf(-0.5f * x * x);
}

template <int ACT>
struct EpBiasGeluSaveBf16 {  // out = act(v + bias), pre = v + bias (kept for the backward); act = GELU or QuickGELU
    bf16_t* out;
    bf16_t* pre_out;
    const float* bias;
    long ldo;
    using Cols = ColsBias;
    __device__ __forceinline__ Cols cols(int n) const { return load_bias(bias, n); }
    __device__ __forceinline__ void operator()(long m, int n, const float* v, const Cols& c) const {
        const float r[4] = {v[0] + c.b.x, v[1] + c.b.y, v[2] + c.b.z, v[3] + c.b.w};
        *reinterpret_cast<uint2*>(pre_out + (size_t)m * ldo + n) = make_uint2(pack2bf(r[0], r[1]), pack2bf(r[2], r[3]));
        *reinterpret_cast<uint2*>(out + (size_t)m * ldo + n) =
            make_uint2(pack2bf(act_fwd(ACT, r[0]), act_fwd(ACT, r[1])), pack2bf(act_fwd(ACT, r[2]), act_fwd(ACT, r[3])));
    }
};

struct ColsNone {};

template <int ACT>
struct EpMulDGeluBf16 {  // out = v * act'(pre)
    bf16_t* out;
    const bf16_t* pre_in;
    long ldo;
    using Cols = ColsNone;
    using Pre = uint2;
    __device__ __forceinline__ Cols cols(int) const { return {}; }
    __device__ __forceinline__ Pre pre(long m, int n) const {
        return *reinterpret_cast<const uint2*>(pre_in + (size_t)m * ldo + n);
    }
    __device__ __forceinline__ void operator()(long m, int n, const float* v, const Cols&, const Pre& raw) const {
        const float x[4] = {bf2f((bf16_t)(raw.x & 0xffff)), bf2f((bf16_t)(raw.x >> 16)), bf2f((bf16_t)(raw.y & 0xffff)),
                            bf2f((bf16_t)(raw.y >> 16))};
        float r[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) r[j] = v[j] * act_grad(ACT, x[j]);
        *reinterpret_cast<uint2*>(out + (size_t)m * ldo + n) = make_uint2(pack2bf(r[0], r[1]), pack2bf(r[2], r[3]));
    }
};

template <int ACT>
struct EpBiasActF32 {  // out[m][n] = act(v + bias[n]) in fp32
    float* out;
    const float* bias;
    long ldo;
    using Cols = ColsBias;
    __device__ __forceinline__ Cols cols(int n) const { return load_bias(bias, n); }
    __device__ __forceinline__ void operator()(long m, int n, const float* v, const Cols& c) const {
        float r[4] = {v[0] + c.b.x, v[1] + c.b.y, v[2] + c.b.z, v[3] + c.b.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (ACT == ACT_RELU) r[j] = fmaxf(r[j], 0.f);
            if (ACT == ACT_GELU) r[j] = gelu_erf(r[j]);
        }
        *reinterpret_cast<float4*>(out + (size_t)m * ldo + n) = make_float4(r[0], r[1], r[2], r[3]);
    }
};

struct EpResidual {  // x[m][n] += gamma[n] * (v + bias[n])   (LayerScale + residual, block.py:92-117)
    float* x;
    const float* bias;
    const float* gamma;  // may be null (no LayerScale)
    long ldx;
    struct Cols {
        float4 b, g;
    };
    using Pre = float4;
    __device__ __forceinline__ Cols cols(int n) const {
        return Cols{bias ? *reinterpret_cast<const float4*>(bias + n) : make_float4(0, 0, 0, 0),
                    gamma ? *reinterpret_cast<const float4*>(gamma + n) : make_float4(1, 1, 1, 1)};
    }
    __device__ __forceinline__ Pre pre(long m, int n) const { return *reinterpret_cast<const float4*>(x + (size_t)m * ldx + n); }
    __device__ __forceinline__ void operator()(long m, int n, const float* v, const Cols& c, const Pre& old) const {
        float4 o = old;
        o.x += c.g.x * (v[0] + c.b.x);
        o.y += c.g.y * (v[1] + c.b.y);
        o.z += c.g.z * (v[2] + c.b.z);
        o.w += c.g.w * (v[3] + c.b.w);
        *reinterpret_cast<float4*>(x + (size_t)m * ldx + n) = o;
    }
};

// EpResidual that also hands the NEXT GEMM what it needs to fold the following LayerNorm in (EpLnFold below): a 16-bit
// copy of the updated fp32 rows and, per row and wave column group, the sum / sum of squares of that copy.  The ViT
// block then has no LayerNorm launch between its GEMMs (block.py:92-117: x + ls(attn(norm1 x)), x + ls(mlp(norm2 x))).
struct EpResidualStats {
    static constexpr bool kRowStats = true;
    float* x;
    const float* bias;
    const float* gamma;  // may be null
    long ldx;
    unsigned short* x16;  // [M][ldx] IEEE half copy of the updated rows
    float* stats;         // [slots][M][2]
    long M;
    using Cols = EpResidual::Cols;
    using Pre = float4;
    __device__ __forceinline__ Cols cols(int n) const {
        return Cols{bias ? *reinterpret_cast<const float4*>(bias + n) : make_float4(0, 0, 0, 0),
                    gamma ? *reinterpret_cast<const float4*>(gamma + n) : make_float4(1, 1, 1, 1)};
    }
    __device__ __forceinline__ Pre pre(long m, int n) const { return *reinterpret_cast<const float4*>(x + (size_t)m * ldx + n); }
    __device__ __forceinline__ void operator()(long m, int n, const float* v, const Cols& c, const Pre& old, float& s1, float& s2) const {
        float4 o = old;
        o.x += c.g.x * (v[0] + c.b.x);
        o.y += c.g.y * (v[1] + c.b.y);
        o.z += c.g.z * (v[2] + c.b.z);
        o.w += c.g.w * (v[3] + c.b.w);
        *reinterpret_cast<float4*>(x + (size_t)m * ldx + n) = o;
        const uint2 q = make_uint2(pack2h_sat(o.x, o.y), pack2h_sat(o.z, o.w));
        *reinterpret_cast<uint2*>(x16 + (size_t)m * ldx + n) = q;
        const float q0 = h_lo(q.x), q1 = h_hi(q.x), q2 = h_lo(q.y), q3 = h_hi(q.y);
        s1 += (q0 + q1) + (q2 + q3);
        s2 += (q0 * q0 + q1 * q1) + (q2 * q2 + q3 * q3);
    }
};

template <bool F16 = false>
struct EpAxpyResBf16 {  // out = res + alpha * (v + bias), bf16 (or half) in/out (FeatUp JBUStack final fix-up, LoftUp residuals)
    bf16_t* out;
    const bf16_t* res;
    const float* bias;
    float alpha;
    long ldo;
    using Cols = ColsBias;
    using Pre = uint2;
    __device__ __forceinline__ Cols cols(int n) const { return load_bias(bias, n); }
    __device__ __forceinline__ Pre pre(long m, int n) const { return *reinterpret_cast<const uint2*>(res + (size_t)m * ldo + n); }
    __device__ __forceinline__ void operator()(long m, int n, const float* v, const Cols& c, const Pre& u) const {
        const float x0 = F16 ? h_lo(u.x) : __uint_as_float(u.x << 16), x1 = F16 ? h_hi(u.x) : __uint_as_float(u.x & 0xffff0000u);
        const float x2 = F16 ? h_lo(u.y) : __uint_as_float(u.y << 16), x3 = F16 ? h_hi(u.y) : __uint_as_float(u.y & 0xffff0000u);
        const float r0 = x0 + alpha * (v[0] + c.b.x), r1 = x1 + alpha * (v[1] + c.b.y);
        const float r2 = x2 + alpha * (v[2] + c.b.z), r3 = x3 + alpha * (v[3] + c.b.w);
        *reinterpret_cast<uint2*>(out + (size_t)m * ldo + n) = make_uint2(pack2o_sat<!F16>(r0, r1), pack2o_sat<!F16>(r2, r3));
    }
};

// LayerNorm folded into the NEXT GEMM (LoftUp's half-precision inference stream, loftup/layers.py:186-228: every
// cross-attention / feed-forward / final projection reads LayerNorm(x)).  With g = LN's gain and b its bias,
//     W LN(x) + c  =  rstd * ( (W diag(g)) x  -  mean * s )  +  (c + W b),      s[n] = sum_k (W diag(g))[n][k],
// so the consumer multiplies the RAW row x by the folded weights and corrects per row in its epilogue: the LayerNorm
// pass over the 1.6 M x 448 pixel map (read + write, 0.6 ms at batch 8, nine per forward = 15 % of the stage) disappears.
// The row statistics come from the PRODUCER of x: EpAxpyResStats is EpAxpyResBf16 that also emits, per output row and
// wave column group, sum(r) and sum(r^2) of what it stores -- partial[slot][m][2], slot = n-tile * WN + n-wave (the
// padded columns of LoftUp's 404 -> 448 channel map are exact zeros and drop out).  No atomics, no extra pass.
template <bool F16 = false>
struct EpAxpyResStats {
    static constexpr bool kRowStats = true;
    bf16_t* out;
    const bf16_t* res;
    const float* bias;
    float alpha;
    long ldo;
    float* stats;  // [slots][M][2]
    long M;
    using Cols = ColsBias;
    using Pre = uint2;
    __device__ __forceinline__ Cols cols(int n) const { return load_bias(bias, n); }
    __device__ __forceinline__ Pre pre(long m, int n) const { return *reinterpret_cast<const uint2*>(res + (size_t)m * ldo + n); }
    __device__ __forceinline__ void operator()(long m, int n, const float* v, const Cols& c, const Pre& u, float& s1, float& s2) const {
        const float x0 = F16 ? h_lo(u.x) : __uint_as_float(u.x << 16), x1 = F16 ? h_hi(u.x) : __uint_as_float(u.x & 0xffff0000u);
        const float x2 = F16 ? h_lo(u.y) : __uint_as_float(u.y << 16), x3 = F16 ? h_hi(u.y) : __uint_as_float(u.y & 0xffff0000u);
        const float r0 = x0 + alpha * (v[0] + c.b.x), r1 = x1 + alpha * (v[1] + c.b.y);
        const float r2 = x2 + alpha * (v[2] + c.b.z), r3 = x3 + alpha * (v[3] + c.b.w);
        const uint2 q = make_uint2(pack2o_sat<!F16>(r0, r1), pack2o_sat<!F16>(r2, r3));
        *reinterpret_cast<uint2*>(out + (size_t)m * ldo + n) = q;
        // statistics of the values AS STORED (what the consumer GEMM multiplies)
        const float q0 = F16 ? h_lo(q.x) : __uint_as_float(q.x << 16), q1 = F16 ? h_hi(q.x) : __uint_as_float(q.x & 0xffff0000u);
        const float q2 = F16 ? h_lo(q.y) : __uint_as_float(q.y << 16), q3 = F16 ? h_hi(q.y) : __uint_as_float(q.y & 0xffff0000u);
        s1 += (q0 + q1) + (q2 + q3);
        s2 += (q0 * q0 + q1 * q1) + (q2 * q2 + q3 * q3);
    }
};

// The consumer side: out = act( rstd[m] * (v - mean[m] * s[n]) + bias[n] ), 16-bit out.
template <int ACT, bool F16 = false>
struct EpLnFold {
    bf16_t* out;
    const float* bias;   // c + W b
    const float* ssum;   // s[n]
    const float* stats;  // [slots][M][2] from the producer
    long M;
    int slots;
    float inv_d, eps;    // 1 / (channels the LayerNorm runs over), its epsilon
    long ldo;
    struct Cols {
        float4 b, s;
    };
    struct RowCtx {
        float mean, rstd;
    };
    __device__ __forceinline__ Cols cols(int n) const {
        return Cols{*reinterpret_cast<const float4*>(bias + n), *reinterpret_cast<const float4*>(ssum + n)};
    }
    __device__ __forceinline__ RowCtx row_begin(long m) const {
        return row_finish(row_partial(m));
    }
    // The four lanes that hold a row (lane >> 4 = 0..3) share the slot loads (slots <= 8: at most two each, no loop, so a
    // caller can issue the loads of all its rows before it needs the first result) and exchange the sums.
    __device__ __forceinline__ float2 row_partial(long m) const {
        const int k = (int)(__lane_id() >> 4);
        float2 a = make_float2(0.f, 0.f), b = a;
        if (k < slots) a = *reinterpret_cast<const float2*>(stats + ((size_t)k * M + m) * 2);
        if (k + 4 < slots) b = *reinterpret_cast<const float2*>(stats + ((size_t)(k + 4) * M + m) * 2);
        return make_float2(a.x + b.x, a.y + b.y);
    }
    __device__ __forceinline__ RowCtx row_finish(float2 p) const {
        float s1 = p.x, s2 = p.y;
        s1 += __shfl_xor(s1, 16), s2 += __shfl_xor(s2, 16);
        s1 += __shfl_xor(s1, 32), s2 += __shfl_xor(s2, 32);
        const float mean = s1 * inv_d;
        return RowCtx{mean, rsqrtf(fmaxf(s2 * inv_d - mean * mean, 0.f) + eps)};
    }
    __device__ __forceinline__ uint2 pack(long, int, const float* v, const Cols& c, const RowCtx& r) const {
        float o[4] = {r.rstd * (v[0] - r.mean * c.s.x) + c.b.x, r.rstd * (v[1] - r.mean * c.s.y) + c.b.y,
                      r.rstd * (v[2] - r.mean * c.s.z) + c.b.z, r.rstd * (v[3] - r.mean * c.s.w) + c.b.w};
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (ACT == ACT_GELU) o[j] = (F16 && ISP_GELU_SIG_F16) ? gelu_sig5(o[j]) : gelu_erf(o[j]);
        return make_uint2(pack2o_sat<!F16>(o[0], o[1]), pack2o_sat<!F16>(o[2], o[3]));
    }
    __device__ __forceinline__ void operator()(long m, int n, const float* v, const Cols& c, const RowCtx& r) const {
        *reinterpret_cast<uint2*>(out + (size_t)m * ldo + n) = pack(m, n, v, c, r);
    }
};

// EpLnFold followed by a LayerNorm over the OUTPUT row (LoftUp's tail, loftup.py:139-149: LayerNorm -> 1x1 conv c -> C ->
// channel LayerNorm): y[n] = rstd (v - mean s[n]) + bias[n], out = (y - mean_y) rstd_y g2[n] + b2[n].  The workgroup's tile
// spans the whole output row (N <= BN), so mean_y / rstd_y come from the accumulators: per-wave partial sums through LDS,
// one barrier, normalise in registers -- the [M, C] map is written once instead of written, read and written again.
template <bool F16 = false>
struct EpLnFoldLayerNorm {
    static constexpr bool kRowLayerNorm = true;
    bf16_t* out;
    const float* bias;   // c + W b
    const float* ssum;   // s[n]
    const float* stats;  // [slots][M][2] from the producer of the input rows
    long M;
    int slots;
    float inv_d, eps;    // input LayerNorm: 1 / channels, epsilon
    long ldo;
    const float* g2;     // output LayerNorm gain / bias over the N output channels
    const float* b2;
    float inv_n, eps2;
    struct Cols {
        float4 b, s, g, h;
    };
    using RowCtx = typename EpLnFold<ACT_NONE, F16>::RowCtx;
    __device__ __forceinline__ Cols cols(int n) const {
        return Cols{*reinterpret_cast<const float4*>(bias + n), *reinterpret_cast<const float4*>(ssum + n),
                    *reinterpret_cast<const float4*>(g2 + n), *reinterpret_cast<const float4*>(b2 + n)};
    }
    __device__ __forceinline__ RowCtx row_begin(long m) const {
        return row_finish(row_partial(m));
    }
    // The four lanes that hold a row (lane >> 4 = 0..3) share the slot loads (slots <= 8: at most two each, no loop, so a
    // caller can issue the loads of all its rows before it needs the first result) and exchange the sums.
    __device__ __forceinline__ float2 row_partial(long m) const {
        const int k = (int)(__lane_id() >> 4);
        float2 a = make_float2(0.f, 0.f), b = a;
        if (k < slots) a = *reinterpret_cast<const float2*>(stats + ((size_t)k * M + m) * 2);
        if (k + 4 < slots) b = *reinterpret_cast<const float2*>(stats + ((size_t)(k + 4) * M + m) * 2);
        return make_float2(a.x + b.x, a.y + b.y);
    }
    __device__ __forceinline__ RowCtx row_finish(float2 p) const {
        float s1 = p.x, s2 = p.y;
        s1 += __shfl_xor(s1, 16), s2 += __shfl_xor(s2, 16);
        s1 += __shfl_xor(s1, 32), s2 += __shfl_xor(s2, 32);
        const float mean = s1 * inv_d;
        return RowCtx{mean, rsqrtf(fmaxf(s2 * inv_d - mean * mean, 0.f) + eps)};
    }
    __device__ __forceinline__ void value(f32x4& v, const Cols& c, const RowCtx& r) const {  // acc -> y, in place
        v[0] = r.rstd * (v[0] - r.mean * c.s.x) + c.b.x, v[1] = r.rstd * (v[1] - r.mean * c.s.y) + c.b.y;
        v[2] = r.rstd * (v[2] - r.mean * c.s.z) + c.b.z, v[3] = r.rstd * (v[3] - r.mean * c.s.w) + c.b.w;
    }
    __device__ __forceinline__ uint2 pack(const f32x4& y, const Cols& c, float mean, float rstd) const {
        return make_uint2(pack2o_sat<!F16>((y[0] - mean) * rstd * c.g.x + c.h.x, (y[1] - mean) * rstd * c.g.y + c.h.y),
                          pack2o_sat<!F16>((y[2] - mean) * rstd * c.g.z + c.h.z, (y[3] - mean) * rstd * c.g.w + c.h.w));
    }
};

// The whole-row epilogue of gemm_tile_kernel for EP::kRowLayerNorm (tiles_n == 1).  red: [WN][BM] float2 of LDS behind the
// operand ring's first bytes, which every wave has finished reading (barrier at entry).
template <class CFG, class EP, class RowFn>
__device__ __forceinline__ void row_layernorm_epilogue(const EP& ep, f32x4 (&acc)[CFG::TM][CFG::TN], RowFn row_of, int wm, int wn,
                                                       int fr, int fq, int N, char* smem) {
    constexpr int TM = CFG::TM, TN = CFG::TN, BM = CFG::BM;
    float2* red = reinterpret_cast<float2*>(smem);
    const int n_base = wn * (TN * 16);
    typename EP::Cols cc[TN];
    int ncol[TN];
#pragma unroll
    for (int ni = 0; ni < TN; ++ni) {
        ncol[ni] = n_base + ni * 16 + fq * 4;
        cc[ni] = ep.cols(ncol[ni] < N ? ncol[ni] : N - 4);
    }
    float2 part[TM];
#pragma unroll
    for (int mi = 0; mi < TM; ++mi) {
        const long m = row_of(wm * (TM * 16) + mi * 16 + fr);
        part[mi] = ep.row_partial(m >= 0 ? m : 0);
    }
    __builtin_amdgcn_s_barrier();  // every wave is done reading the operand ring
#pragma unroll
    for (int mi = 0; mi < TM; ++mi) {
        const int rl = wm * (TM * 16) + mi * 16 + fr;
        const long m = row_of(rl);
        const auto ctx = ep.row_finish(part[mi]);
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int ni = 0; ni < TN; ++ni) {
            ep.value(acc[mi][ni], cc[ni], ctx);
            if (ncol[ni] < N) {
                s1 += (acc[mi][ni][0] + acc[mi][ni][1]) + (acc[mi][ni][2] + acc[mi][ni][3]);
                s2 += (acc[mi][ni][0] * acc[mi][ni][0] + acc[mi][ni][1] * acc[mi][ni][1]) +
                      (acc[mi][ni][2] * acc[mi][ni][2] + acc[mi][ni][3] * acc[mi][ni][3]);
            }
        }
        s1 += __shfl_xor(s1, 16), s2 += __shfl_xor(s2, 16);
        s1 += __shfl_xor(s1, 32), s2 += __shfl_xor(s2, 32);
        if (fq == 0) red[wn * BM + rl] = make_float2(s1, s2);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int mi = 0; mi < TM; ++mi) {
        const int rl = wm * (TM * 16) + mi * 16 + fr;
        const long m = row_of(rl);
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int w = 0; w < CFG::WN; ++w) {
            const float2 p = red[w * BM + rl];
            s1 += p.x, s2 += p.y;
        }
        const float mean = s1 * ep.inv_n;
        const float rstd = rsqrtf(fmaxf(s2 * ep.inv_n - mean * mean, 0.f) + ep.eps2);
        if (m < 0) continue;
#pragma unroll
        for (int ni = 0; ni < TN; ++ni)
            if (ncol[ni] < N)
                *reinterpret_cast<uint2*>(ep.out + (size_t)m * ep.ldo + ncol[ni]) = ep.pack(acc[mi][ni], cc[ni], mean, rstd);
    }
}

// relu(v + bias[n] - sum over the 3x3 taps that fall OUTSIDE the image of taps[t][n]), bf16 out.
// Used when a per-pixel affine map z = (I + aW)x + a*b in front of a zero-padded 3x3 conv is folded
// into the conv weights: the constant part a*b only contributes through taps inside the image.
// `bias` already holds b_conv + sum_t taps[t]; border pixels subtract their missing taps.
template <bool F16 = false>
struct EpBiasTapsReluBf16 {
    bf16_t* out;
    const float* bias;
    const float* taps;  // [9][ldo]
    int H, W;
    long ldo;
    using Cols = ColsBias;
    __device__ __forceinline__ Cols cols(int n) const { return load_bias(bias, n); }
    // row context: 9-bit mask of the taps that fall outside the image for output pixel m
    __device__ __forceinline__ unsigned row_begin(long m) const {
        const unsigned rem = (unsigned)m % ((unsigned)H * (unsigned)W);
        const int y = (int)(rem / (unsigned)W), x = (int)(rem % (unsigned)W);
        unsigned mask = 0;
        if (y == 0) mask |= 0x007u;
        if (y == H - 1) mask |= 0x1c0u;
        if (x == 0) mask |= 0x049u;
        if (x == W - 1) mask |= 0x124u;
        return mask;
    }
    __device__ __forceinline__ uint2 pack(long, int n, const float* v, const Cols& c, unsigned outside) const {
        float4 b = c.b;
        if (outside) {
#pragma unroll 1
            for (int t = 0; t < 9; ++t) {
                if ((outside >> t) & 1u) {
                    const float4 tv = *reinterpret_cast<const float4*>(taps + (size_t)t * ldo + n);
                    b.x -= tv.x, b.y -= tv.y, b.z -= tv.z, b.w -= tv.w;
                }
            }
        }
        return make_uint2(pack2o_sat<!F16>(fmaxf(v[0] + b.x, 0.f), fmaxf(v[1] + b.y, 0.f)),
                          pack2o_sat<!F16>(fmaxf(v[2] + b.z, 0.f), fmaxf(v[3] + b.w, 0.f)));
    }
    __device__ __forceinline__ void operator()(long m, int n, const float* v, const Cols& c, unsigned outside) const {
        *reinterpret_cast<uint2*>(out + (size_t)m * ldo + n) = pack(m, n, v, c, outside);
    }
};

// Second head conv fused with the 1x1 classifier (heads/base_head.py:15): the conv output is never
// stored; each wave reduces relu(v + bias) . wcls over its own output channels and writes one fp32
// partial per row: partial[slot][m], slot = n-tile * WN + n-wave.  A tiny kernel sums the slots.
struct EpReluDotPartial {
    static constexpr bool kRowReduce = true;
    float* partial;      // [slots][M]
    const float* bias;   // [N]
    const float* wcls;   // [N]
    long M;
    struct Cols {
        float4 b, w;
    };
    __device__ __forceinline__ Cols cols(int n) const {
        return Cols{*reinterpret_cast<const float4*>(bias + n), *reinterpret_cast<const float4*>(wcls + n)};
    }
    __device__ __forceinline__ float term(int, const float* v, const Cols& c) const {
        return fmaxf(v[0] + c.b.x, 0.f) * c.w.x + fmaxf(v[1] + c.b.y, 0.f) * c.w.y + fmaxf(v[2] + c.b.z, 0.f) * c.w.z +
               fmaxf(v[3] + c.b.w, 0.f) * c.w.w;
    }
};

struct EpTokens {  // patch-embed: token row b*(T+1)+1+t gets v + bias[n] + pos[1+t][n]
    float* x;
    const float* bias;  // b_img + b_click, pre-summed
    const float* pos;   // [T+1][N] interpolated pos-embed (row 0 = cls), may be null
    int T;
    long ldx;
    using Cols = ColsBias;
    using Pre = float4;
    __device__ __forceinline__ Cols cols(int n) const { return load_bias(bias, n); }
    __device__ __forceinline__ Pre pre(long m, int n) const {
        const int t = (int)(m % T);
        return pos ? *reinterpret_cast<const float4*>(pos + (size_t)(1 + t) * ldx + n) : make_float4(0, 0, 0, 0);
    }
    __device__ __forceinline__ void operator()(long m, int n, const float* v, const Cols& c, const Pre& pp) const {
        const long b = m / T;
        const int t = (int)(m - b * T);
        *reinterpret_cast<float4*>(x + (size_t)(b * (T + 1) + 1 + t) * ldx + n) =
            make_float4(v[0] + c.b.x + pp.x, v[1] + c.b.y + pp.y, v[2] + c.b.z + pp.z, v[3] + c.b.w + pp.w);
    }
};

// ------------------------------------------------------------------------------ the engine
__device__ __forceinline__ int swz(int row, int chunk) { return chunk ^ ((row >> 1) & 7); }

// Shared epilogue driver.  A lane holds, for accumulator tile (mi, ni), output row (row_base + 16*mi +
// fr) and the 4 consecutive columns n_base + 16*ni + 4*fq + j.  `row_of(local_row)` maps a tile row
// to the global output row (or -1).  EP::kRowReduce epilogues reduce over the wave's columns instead
// of storing them (conv + classifier fusion): slot = which (n-tile, n-wave) partial this wave owns.
// FULL = the wave-uniform fast path for tiles that lie entirely inside the matrix: no per-lane row / column
// checks, hence no divergent branches -- behind those the compiler has to place s_waitcnt vmcnt(0) at every join,
// which makes each store wait for the previous store's acknowledgement (measured: 11 us of a 66 us conv tile).
template <int TM, int TN, bool FULL = false, class EP, class RowFn>
__device__ __forceinline__ void run_epilogue(const EP& ep, f32x4 (&acc)[TM][TN], RowFn row_of, int row_base, int fr,
                                             int fq, int n_base, int N, int slot) {
    // per-column constants of the wave's TN column groups (columns past N read the last valid group: never used)
    typename EP::Cols cc[TN];
    int ncol[TN];
#pragma unroll
    for (int ni = 0; ni < TN; ++ni) {
        ncol[ni] = n_base + ni * 16 + fq * 4;
        cc[ni] = ep.cols(FULL || ncol[ni] < N ? ncol[ni] : N - 4);
    }
    if constexpr (requires { EP::kRowReduce; }) {
#pragma unroll
        for (int mi = 0; mi < TM; ++mi) {
            const long m = row_of(row_base + mi * 16 + fr);
            float sum = 0.f;
#pragma unroll
            for (int ni = 0; ni < TN; ++ni) {
                const float v[4] = {acc[mi][ni][0], acc[mi][ni][1], acc[mi][ni][2], acc[mi][ni][3]};
                if (FULL || ncol[ni] < N) sum += ep.term(ncol[ni], v, cc[ni]);
            }
            sum += __shfl_xor(sum, 16);
            sum += __shfl_xor(sum, 32);
            if (fq == 0 && (FULL || m >= 0)) ep.partial[(size_t)slot * ep.M + m] = sum;
        }
    } else {
#pragma unroll
        for (int mi = 0; mi < TM; ++mi) {
            const long m = row_of(row_base + mi * 16 + fr);
            if constexpr (requires { ep.pre(m, 0); }) {
                // all of the row's operand loads first (rows / columns outside the matrix read element (0, N-4))
                typename EP::Pre pp[TN];
#pragma unroll
                for (int ni = 0; ni < TN; ++ni)
                    pp[ni] = ep.pre(FULL || m >= 0 ? m : 0, FULL || ncol[ni] < N ? ncol[ni] : N - 4);
                if constexpr (requires { EP::kRowStats; }) {
                    // stores + per-row sum / sum of squares of the stored values over the wave's columns; the four lanes
                    // that share a row (fq = 0..3) reduce, lane fq == 0 writes partial[slot][m]
                    float s1 = 0.f, s2 = 0.f;
                    const bool row_ok = FULL || m >= 0;
#pragma unroll
                    for (int ni = 0; ni < TN; ++ni) {
                        if (!row_ok || (!FULL && ncol[ni] >= N)) continue;
                        const float v[4] = {acc[mi][ni][0], acc[mi][ni][1], acc[mi][ni][2], acc[mi][ni][3]};
                        ep(m, ncol[ni], v, cc[ni], pp[ni], s1, s2);
                    }
                    s1 += __shfl_xor(s1, 16), s2 += __shfl_xor(s2, 16);
                    s1 += __shfl_xor(s1, 32), s2 += __shfl_xor(s2, 32);
                    if (fq == 0 && row_ok) *reinterpret_cast<float2*>(ep.stats + ((size_t)slot * ep.M + m) * 2) = make_float2(s1, s2);
                    continue;
                }
                if (!FULL && m < 0) continue;
#pragma unroll
                for (int ni = 0; ni < TN; ++ni) {
                    if (!FULL && ncol[ni] >= N) continue;
                    const float v[4] = {acc[mi][ni][0], acc[mi][ni][1], acc[mi][ni][2], acc[mi][ni][3]};
                    if constexpr (!requires { EP::kRowStats; }) ep(m, ncol[ni], v, cc[ni], pp[ni]);
                }
            } else if constexpr (requires { EP::kRowStats; }) {  // (no per-element operand: EpBiasActStats)
                float s1 = 0.f, s2 = 0.f;
                const bool row_ok = FULL || m >= 0;
#pragma unroll
                for (int ni = 0; ni < TN; ++ni) {
                    if (!row_ok || (!FULL && ncol[ni] >= N)) continue;
                    const float v[4] = {acc[mi][ni][0], acc[mi][ni][1], acc[mi][ni][2], acc[mi][ni][3]};
                    const uint2 q = ep.pack(m, ncol[ni], v, cc[ni]);
                    *reinterpret_cast<uint2*>(ep.out + (size_t)m * ep.ldo + ncol[ni]) = q;
                    EP::add(q, s1, s2);
                }
                s1 += __shfl_xor(s1, 16), s2 += __shfl_xor(s2, 16);
                s1 += __shfl_xor(s1, 32), s2 += __shfl_xor(s2, 32);
                if (fq == 0 && row_ok) ep.put_stats(m, slot, s1, s2);
            } else {
                if (!FULL && m < 0) continue;
                if constexpr (requires { ep.row_begin(m); }) {
                    const auto ctx = ep.row_begin(m);
#pragma unroll
                    for (int ni = 0; ni < TN; ++ni) {
                        if (!FULL && ncol[ni] >= N) continue;
                        const float v[4] = {acc[mi][ni][0], acc[mi][ni][1], acc[mi][ni][2], acc[mi][ni][3]};
                        ep(m, ncol[ni], v, cc[ni], ctx);
                    }
                } else {
#pragma unroll
                    for (int ni = 0; ni < TN; ++ni) {
                        if (!FULL && ncol[ni] >= N) continue;
                        const float v[4] = {acc[mi][ni][0], acc[mi][ni][1], acc[mi][ni][2], acc[mi][ni][3]};
                        ep(m, ncol[ni], v, cc[ni]);
                    }
                }
            }
        }
    }
}

// bf16-output epilogues that can hand back their packed values (pack(), no per-element operand, no row reduction)
// may leave through LDS: the accumulator layout gives a lane 4 columns of ONE row, so a direct store instruction is
// 64 separate 8-byte segments (measured: the epilogue was ~40 % of a K = 384..512 GEMM and half of the conv's 9 %).
// The wave's TM*16 x TN*16 tile is written to a wave-private LDS tile (pitch TN*32 + 16 bytes: <= 2-way conflicts)
// and read back as 16 bytes per lane with neighbouring lanes contiguous along the row.
template <class EP>
constexpr bool kStagedStore = requires(const EP& e) { e.out; e.ldo; typename EP::Cols; } &&
                              !requires { EP::kRowReduce; } && !requires { typename EP::Pre; } &&
                              (requires(const EP& e, const float* v, const typename EP::Cols& c) { e.pack(0L, 0, v, c); } ||
                               requires(const EP& e, const float* v, const typename EP::Cols& c) { e.pack(0L, 0, v, c, 0u); } ||
                               requires { typename EP::RowCtx; });
template <int TM, int TN>
constexpr int kStageBytes = TM * 16 * (TN * 32 + 16);  // per wave

template <class EP>
__device__ __forceinline__ bool staged_store_ok(const EP& ep) {  // 16-byte chunks need 8-element alignment
    if constexpr (kStagedStore<EP>) return ep.ldo % 8 == 0 && (reinterpret_cast<size_t>(ep.out) & 15) == 0;
    else return false;
}

// row_in(r): global output row of tile row r (the tile is interior: every row and column exists); stg: this wave's
// kStageBytes<TM, TN> of LDS that no other wave touches any more.
template <int TM, int TN, class EP, class RowFn>
__device__ __forceinline__ void staged_epilogue(const EP& ep, f32x4 (&acc)[TM][TN], RowFn row_in, int row_base, int fr,
                                                int fq, int lane, int n_base, char* stg, int slot = 0) {
    constexpr int SP = TN * 32 + 16, CPR = TN * 2;  // row pitch; 16-byte chunks per row
    static_assert((TM * 16 * CPR) % 64 == 0);
    typename EP::Cols cc[TN];
#pragma unroll
    for (int ni = 0; ni < TN; ++ni) cc[ni] = ep.cols(n_base + ni * 16 + fq * 4);
    [[maybe_unused]] float2 part[TM];
    if constexpr (requires { ep.row_partial(0L); }) {  // all rows' statistics loads in flight before the first is needed
#pragma unroll
        for (int mi = 0; mi < TM; ++mi) part[mi] = ep.row_partial(row_in(row_base + mi * 16 + fr));
    }
#pragma unroll
    for (int mi = 0; mi < TM; ++mi) {
        [[maybe_unused]] const long m = row_in(row_base + mi * 16 + fr);
        auto put = [&](int ni, const uint2& o) { *reinterpret_cast<uint2*>(stg + (mi * 16 + fr) * SP + ni * 32 + fq * 8) = o; };
        if constexpr (requires { ep.row_begin(m); }) {
            const auto ctx = [&] {
                if constexpr (requires { ep.row_partial(0L); }) return ep.row_finish(part[mi]);
                else return ep.row_begin(m);
            }();
#pragma unroll
            for (int ni = 0; ni < TN; ++ni) {
                const float v[4] = {acc[mi][ni][0], acc[mi][ni][1], acc[mi][ni][2], acc[mi][ni][3]};
                put(ni, ep.pack(m, n_base + ni * 16 + fq * 4, v, cc[ni], ctx));
            }
        } else if constexpr (requires { EP::kRowStats; }) {
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int ni = 0; ni < TN; ++ni) {
                const float v[4] = {acc[mi][ni][0], acc[mi][ni][1], acc[mi][ni][2], acc[mi][ni][3]};
                const uint2 q = ep.pack(m, n_base + ni * 16 + fq * 4, v, cc[ni]);
                put(ni, q);
                EP::add(q, s1, s2);
            }
            s1 += __shfl_xor(s1, 16), s2 += __shfl_xor(s2, 16);
            s1 += __shfl_xor(s1, 32), s2 += __shfl_xor(s2, 32);
            if (fq == 0) ep.put_stats(m, slot, s1, s2);
        } else {
#pragma unroll
            for (int ni = 0; ni < TN; ++ni) {
                const float v[4] = {acc[mi][ni][0], acc[mi][ni][1], acc[mi][ni][2], acc[mi][ni][3]};
                put(ni, ep.pack(m, n_base + ni * 16 + fq * 4, v, cc[ni]));
            }
        }
    }
    // (written and read by the same wave: the compiler's lgkmcnt wait orders them)
#pragma unroll
    for (int j = 0; j < TM * 16 * CPR / 64; ++j) {
        const int id = j * 64 + lane, row = id / CPR, c = id - row * CPR;
        const uint4 q = *reinterpret_cast<const uint4*>(stg + row * SP + c * 16);
        *reinterpret_cast<uint4*>(ep.out + (size_t)row_in(row_base + row) * ep.ldo + n_base + c * 8) = q;
    }
}

template <class CFG, class AL, class EP, bool F16 = false>  // F16: IEEE-half operands (isp_gemm_f16), same MFMA rate
__global__ __launch_bounds__(CFG::THREADS, CFG::MINW) void gemm_tile_kernel(AL al, const bf16_t* __restrict__ Wt, long M,
                                                                          int N, int K, int tiles_n, int nwg, EP ep) {
    constexpr int BM = CFG::BM, BN = CFG::BN, NW = CFG::NW, TM = CFG::TM, TN = CFG::TN, PA = CFG::PA, PW = CFG::PW;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wg = xcd_remap(blockIdx.x, nwg);
    const int tn = wg % tiles_n;
    const long tm = wg / tiles_n;
    const int n0 = tn * BN;

    // --- staging assignment: wave `wid` issues DMA pieces wid, wid+NW, ... of each operand tile;
    // piece q covers tile rows 8q..8q+7 (1 KiB).  Lane -> (row = 8q + lane/8, phys chunk =
    // lane%8), source chunk = swz(row, phys).
    const int lrow = lane >> 3, pchunk = lane & 7;
    const bf16_t* w_src[PW];
#pragma unroll
    for (int i = 0; i < PA; ++i) {
        const int row = (wid + i * NW) * 8 + lrow;
        al.template init<BM>(i, tm, row, swz(row, pchunk));
    }
#pragma unroll
    for (int i = 0; i < PW; ++i) {
        const int row = (wid + i * NW) * 8 + lrow;
        const int n = n0 + row;
        w_src[i] = Wt + (size_t)(n < N ? n : N - 1) * K + swz(row, pchunk) * 8;
    }
    auto stage = [&](char* buf) {  // issues the DMA of the NEXT un-staged K-step and moves the cursors on
#pragma unroll
        for (int i = 0; i < PA; ++i) glds16(al.ptr(i), buf + (wid + i * NW) * 1024);
#pragma unroll
        for (int i = 0; i < PW; ++i) glds16(w_src[i], buf + CFG::A_BYTES + (wid + i * NW) * 1024);
        const long wstep = al.advance();
#pragma unroll
        for (int i = 0; i < PW; ++i) w_src[i] += wstep;
    };

    // --- fragment read addresses (bytes inside a stage)
    const int wm = wid / CFG::WN, wn = wid % CFG::WN;
    const int fr = lane & 15, fq = lane >> 4;
    int a_off[TM][2], w_off[TN][2];
#pragma unroll
    for (int t = 0; t < TM; ++t) {
        const int ra = wm * (TM * 16) + t * 16 + fr;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) a_off[t][ks] = ra * 128 + swz(ra, ks * 4 + fq) * 16;
    }
#pragma unroll
    for (int t = 0; t < TN; ++t) {
        const int rw = wn * (TN * 16) + t * 16 + fr;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) w_off[t][ks] = CFG::A_BYTES + rw * 128 + swz(rw, ks * 4 + fq) * 16;
    }

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto compute = [&](const char* buf) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 fa[TM], fw[TN];
#pragma unroll
            for (int t = 0; t < TM; ++t) fa[t] = *reinterpret_cast<const bf16x8*>(buf + a_off[t][ks]);
#pragma unroll
            for (int t = 0; t < TN; ++t) fw[t] = *reinterpret_cast<const bf16x8*>(buf + w_off[t][ks]);
#pragma unroll
            for (int mi = 0; mi < TM; ++mi)
#pragma unroll
                for (int ni = 0; ni < TN; ++ni)
                    if constexpr (F16)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, fw[ni]),
                                                                            __builtin_bit_cast(f16x8_t, fa[mi]), acc[mi][ni], 0, 0, 0);
                    else
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[ni], fa[mi], acc[mi][ni], 0, 0, 0);
        }
    };

    // --- K loop: NST-deep LDS ring, ONE barrier per K-step, LDS-DMA kept in flight across it.
    //   step t:  wait until only the newest (NST-2) stages of this wave's DMA are outstanding
    //            (=> stage t has landed), retire own LDS reads, barrier (=> everyone's pieces of
    //            stage t landed AND everyone finished reading stage t-1, whose buffer is the one
    //            refilled next), issue the DMA of stage t+NST-1, then MFMA on stage t.
    // Raw s_barrier + counted vmcnt: __syncthreads() would drain the DMA (vmcnt(0)) every step.
    const int nk = K / BK;
    constexpr int NST = CFG::NST, PIECES = PA + PW;
#pragma unroll
    for (int s0 = 0; s0 < NST - 1; ++s0)
        if (s0 < nk) stage(smem + s0 * CFG::STAGE);
    auto step = [&](int t, char* cur, char* refill) {
        if (NST == 3 && t + 1 < nk)
            asm volatile("s_waitcnt vmcnt(%0)" ::"i"(PIECES) : "memory");
        else
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
#ifdef ISP_ABLATE_NO_DMA  // timing experiment only: MFMA + fragment reads on stale LDS
        if (t + NST - 1 < nk) (void)al.advance();
#else
        if (t + NST - 1 < nk) stage(refill);
#endif
#ifndef ISP_ABLATE_NO_MFMA  // timing experiment only: DMA + barriers, no fragment reads / MFMA
        compute(cur);
#endif
    };
    int t = 0;
    if constexpr (NST == 2) {
        for (; t + 2 <= nk; t += 2) {  // unrolled by the ring depth: LDS buffers are compile-time constants
            step(t, smem, smem + CFG::STAGE);
            step(t + 1, smem + CFG::STAGE, smem);
        }
        if (t < nk) step(t, smem, smem + CFG::STAGE);
    } else {
        for (; t + 3 <= nk; t += 3) {
            step(t, smem, smem + 2 * CFG::STAGE);
            step(t + 1, smem + CFG::STAGE, smem);
            step(t + 2, smem + 2 * CFG::STAGE, smem + CFG::STAGE);
        }
        if (t < nk) step(t, smem, smem + 2 * CFG::STAGE);
        if (t + 1 < nk) step(t + 1, smem + CFG::STAGE, smem);
    }

    auto row_of = [&](int r) { return al.template out_row<BM>(tm, r); };
#ifdef ISP_ABLATE_GEMM_NO_EPILOGUE  // timing experiment only
    if (acc[0][0][0] != 12345.678f) return;
#endif
    if constexpr (requires { EP::kRowLayerNorm; }) {  // (the launcher guarantees tiles_n == 1)
        row_layernorm_epilogue<CFG>(ep, acc, row_of, wm, wn, fr, fq, N, smem);
    } else {
        // interior tile (wave-uniform): every row of the tile maps to an output row and every column is < N
        if (n0 + BN <= N && al.template tile_full<BM>(tm)) {
            if constexpr (kStagedStore<EP> && NW * kStageBytes<TM, TN> <= CFG::LDS) {
                if (staged_store_ok(ep)) {
                    __builtin_amdgcn_s_barrier();  // every wave is done reading the operand ring
                    staged_epilogue<TM, TN>(ep, acc, row_of, wm * (TM * 16), fr, fq, lane, n0 + wn * (TN * 16),
                                            smem + wid * kStageBytes<TM, TN>);
                    return;
                }
            }
            run_epilogue<TM, TN, true>(ep, acc, row_of, wm * (TM * 16), fr, fq, n0 + wn * (TN * 16), N, tn * CFG::WN + wn);
        } else {
            run_epilogue<TM, TN>(ep, acc, row_of, wm * (TM * 16), fr, fq, n0 + wn * (TN * 16), N, tn * CFG::WN + wn);
        }
    }
}

template <class CFG, bool F16 = false, class AL, class EP>
int launch_gemm(AL al, const void* Wt, long M, int N, int K, EP ep, hipStream_t s) {
    if (M <= 0 || N <= 0 || K <= 0 || K % BK != 0 || N % 4 != 0) return ISP_ERR_INVALID;
    const long tiles_m = al.template num_tiles<CFG::BM>();
    const int tiles_n = (N + CFG::BN - 1) / CFG::BN;
    const long nwg = tiles_m * tiles_n;
    if (nwg > 0x7fffffffL) return ISP_ERR_INVALID;
    static bool attr_done = false;  // per instantiation; raising the dynamic-LDS cap is idempotent
    auto kern = gemm_tile_kernel<CFG, AL, EP, F16>;
    if (!attr_done) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, CFG::LDS) != hipSuccess)
            return ISP_ERR_LAUNCH;
        attr_done = true;
    }
    kern<<<(unsigned)nwg, CFG::THREADS, CFG::LDS, s>>>(al, (const bf16_t*)Wt, M, N, K, tiles_n, (int)nwg, ep);
    return isp_launch_status();
}

// ------------------------------------------------------------------------------ patch conv
// 3x3 / stride 1 / pad 1 convolution with the INPUT PATCH resident in LDS.  The tile engine above
// stages a fresh 256 x 64 activation tile for every one of the 9 taps although the 9 tiles are shifted
// views of one 18 x 18 x 64 input patch; here that patch (41 KiB, zero outside the image) is staged
// ONCE per 64-channel block and the A fragments of each tap are read from it at shifted addresses.
// LDS-DMA traffic per channel block drops from 9 x (32 + 24) KiB to 41 + 9 x 24 KiB (0.51x): measured
// (B=8, 448^2, C=N=384) the staging path alone took 2.95 ms and the MFMA path alone 2.88 ms of a
// 3.73 ms launch, i.e. the kernel was LDS-DMA bound as much as MFMA bound.
//   block = 16 x 16 output pixels x 192 output channels, 8 waves (4 pixel-row groups x 2 channel
//   halves), K order = channel block major, tap minor; weight tiles in a 2-stage ring (one
//   vmcnt(0)+barrier per K-step; a 3-stage ring fetched two steps ahead measured 6 % slower), the next
//   channel block's patch trickles in one piece per wave per K-step into a second patch buffer.
//   LDS = 2 x 42 KiB + 2 x 24 KiB = 132 KiB.
constexpr int PT = 16, PW_ = 18, PPIX = PW_ * PW_;       // 16x16 outputs, 18x18 inputs
constexpr int P_PIECES = (PPIX + 7) / 8 + 1;              // 42 one-KiB pieces (8 pixels x 128 B)
constexpr int P_BYTES = P_PIECES * 1024;
constexpr int P_WST = 2;  // weight ring depth (a 3-stage ring fetched two K-steps ahead measured the same)

// TN = 16-channel tiles per wave: 6 -> 192 output channels per block (C = N = 384 heads), 8 -> 256 (N = 1024).
template <class EP, int TN>
__global__ __launch_bounds__(512, 2) void conv3x3_patch_kernel(const bf16_t* __restrict__ in,
                                                               const bf16_t* __restrict__ Wt, int H, int W, int C,
                                                               int N, int tiles_x, int tiles_y, int tiles_n, int nwg,
                                                               EP ep) {
    constexpr int TM = 4, NWV = 8, PPW = (P_PIECES + NWV - 1) / NWV;  // 6 patch pieces per wave
    constexpr int PBN = 2 * TN * 16, PWB = PBN * BK * 2, WPW = PBN / 8 / NWV;  // weight pieces per wave (3 or 4)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wg = xcd_remap(blockIdx.x, nwg);
    const int tn = wg % tiles_n;
    const int tmi = wg / tiles_n;
    const int per_img = tiles_x * tiles_y;
    const int b = tmi / per_img, tt = tmi - b * per_img;
    const int y0 = (tt / tiles_x) * PT, x0 = (tt % tiles_x) * PT, n0 = tn * PBN;
    const long K = 9L * C;
    const int cblocks = C / BK;

    // --- patch DMA slots: wave w owns pieces w, w+8, ...; lane -> (pixel = 8*piece + lane/8, phys chunk).
    // 32-bit element offsets from the image base + a validity mask keep this at 7 registers.
    const bf16_t* const img = in + (size_t)b * H * W * C;
    const bf16_t* const zero = reinterpret_cast<const bf16_t*>(g_isp_zero16);
    unsigned poff[PPW], pmask = 0;
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        const int pix = (wid + NWV * i) * 8 + (lane >> 3);
        const int py = pix / PW_, px = pix - py * PW_;
        const int iy = y0 - 1 + py, ix = x0 - 1 + px;
        const bool ok = pix < PPIX && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
        poff[i] = ok ? (unsigned)((iy * W + ix) * C + swz(pix, lane & 7) * 8) : 0u;
        pmask |= ok ? 1u << i : 0u;
    }
    auto issue_patch = [&](int i, int cb, char* buf) {
        if (wid + NWV * i < P_PIECES)
            glds16((pmask >> i & 1) ? img + poff[i] + cb * BK : zero, buf + (wid + NWV * i) * 1024);
    };
    // --- weight DMA slots: PBN/8 pieces per stage, WPW per wave
    const bf16_t* const wbase = Wt + (size_t)n0 * K;
    unsigned woff[WPW];
#pragma unroll
    for (int i = 0; i < WPW; ++i) {
        const int row = (wid + NWV * i) * 8 + (lane >> 3);
        woff[i] = (unsigned)((n0 + row < N ? row : N - 1 - n0) * K + swz(row, lane & 7) * 8);
    }
    auto issue_w = [&](long col, char* buf) {
#pragma unroll
        for (int i = 0; i < WPW; ++i) glds16(wbase + col + woff[i], buf + (wid + NWV * i) * 1024);
    };

    // --- fragment geometry
    const int wm = wid >> 1, wn = wid & 1;
    const int fr = lane & 15, fq = lane >> 4;
    const int p00 = (4 * wm + 1) * PW_ + fr + 1;  // patch pixel of fragment row fr of the wave's first pixel row
    // weight rows wn*96 + 16*t + fr: the swizzle term (row>>1)&7 depends on fr only
    const int w_off0 = (wn * (TN * 16) + fr) * 128 + swz(fr, fq) * 16;
    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    char* const s_w = smem + 2 * P_BYTES;
    // --- prologue: whole patch of channel block 0, weight tile of step 0
#pragma unroll
    for (int i = 0; i < PPW; ++i) issue_patch(i, 0, smem);
    issue_w(0, s_w);

    // A fragments of K-half 0 of the NEXT K-step are read before that step's barrier (the patch is
    // already resident; only the weight tile needs the barrier), so that after the barrier the first
    // MFMA waits for one weight read instead of ten reads: all 8 waves leave the barrier in phase and
    // nobody covers that bubble.
    auto read_a = [&](bf16x8 (&fa)[TM], const char* patch, int shift, int ks) {
#pragma unroll
        for (int t = 0; t < TM; ++t) {
            const int p = p00 + t * PW_ + shift;
            fa[t] = *reinterpret_cast<const bf16x8*>(patch + p * 128 + swz(p, ks * 4 + fq) * 16);
        }
    };
    // weight fragments are read (and consumed) in NH groups of TNH tiles: with TN = 8 a single group would not
    // fit beside the 128 accumulator registers
    constexpr int NH = TN > 6 ? 2 : 1, TNH = TN / NH;
    auto read_w = [&](bf16x8 (&fw)[TNH], const char* wb, int ks, int half) {
#pragma unroll
        for (int t = 0; t < TNH; ++t)
            fw[t] = *reinterpret_cast<const bf16x8*>(wb + ((w_off0 ^ (ks * 64)) + (half * TNH + t) * 2048));
    };
    auto mma = [&](const bf16x8 (&fa)[TM], const bf16x8 (&fw)[TNH], int half) {
#pragma unroll
        for (int mi = 0; mi < TM; ++mi)
#pragma unroll
            for (int ni = 0; ni < TNH; ++ni)
                acc[mi][half * TNH + ni] =
                    __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[ni], fa[mi], acc[mi][half * TNH + ni], 0, 0, 0);
    };
    auto tap_shift = [&](int tap) {
        int shift = (tap / 3 - 1) * PW_ + (tap % 3 - 1);
        asm volatile("" : "+s"(shift));  // opaque: keeps the 72 per-tap fragment addresses out of registers
        return shift;
    };

    const int nsteps = 9 * cblocks;
    bf16x8 fa0[TM];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    read_a(fa0, smem, tap_shift(0), 0);
    int s = 0;
    for (int cb = 0; cb < cblocks; ++cb) {
        const char* patch = smem + (cb & 1) * P_BYTES;
        char* patch_next = smem + ((cb + 1) & 1) * P_BYTES;
        const bool more = cb + 1 < cblocks;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap, ++s) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifndef ISP_ABLATE_NO_BARRIER
            __builtin_amdgcn_s_barrier();
#endif
#ifndef ISP_ABLATE_NO_DMA  // timing experiments only
            if (tap < PPW && more) issue_patch(tap, cb + 1, patch_next);  // next patch: one piece / wave / step
            if (s + 1 < nsteps) {  // weight tile of the next K-step: (cb, tap+1) or (cb+1, 0)
                const long col = tap < 8 ? (long)(tap + 1) * C + (long)cb * BK : (long)(cb + 1) * BK;
                issue_w(col, s_w + ((s + 1) & 1) * PWB);
            }
#endif
#ifdef ISP_ABLATE_NO_MFMA
            continue;
#endif
            const char* wb = s_w + (s & 1) * PWB;
            bf16x8 fa1[TM], fw[TNH];
            read_w(fw, wb, 0, 0);
            read_a(fa1, patch, tap_shift(tap), 1);
            mma(fa0, fw, 0);
#pragma unroll
            for (int half = 1; half < NH; ++half) {
                read_w(fw, wb, 0, half);
                mma(fa0, fw, half);
            }
#pragma unroll
            for (int half = 0; half < NH; ++half) {
                read_w(fw, wb, 1, half);
                mma(fa1, fw, half);
            }
            read_a(fa0, tap < 8 ? patch : patch_next, tap_shift(tap < 8 ? tap + 1 : 0), 0);
        }
    }
    auto row_of = [&](int r) -> long {
        const int y = y0 + (r >> 4), x = x0 + (r & 15);
        return (y < H && x < W) ? ((long)b * H + y) * W + x : -1;
    };
#ifdef ISP_ABLATE_NO_EPILOGUE
    if (acc[0][0][0] == 12345.678f)
#endif
    {
        if (y0 + PT <= H && x0 + PT <= W && n0 + PBN <= N) {  // interior tile: no per-lane checks
            auto row_in = [&](int r) -> long { return ((long)b * H + y0 + (r >> 4)) * W + x0 + (r & 15); };
            if constexpr (kStagedStore<EP>) {
                if (staged_store_ok(ep)) {
                    __builtin_amdgcn_s_barrier();  // every wave is done with the patch and the weight ring
                    staged_epilogue<TM, TN>(ep, acc, row_in, wm * (TM * 16), fr, fq, lane, n0 + wn * (TN * 16),
                                            smem + wid * kStageBytes<TM, TN>);
                    return;
                }
            }
            run_epilogue<TM, TN, true>(ep, acc, row_in, wm * (TM * 16), fr, fq, n0 + wn * (TN * 16), N, tn * 2 + wn);
        } else {
            run_epilogue<TM, TN>(ep, acc, row_of, wm * (TM * 16), fr, fq, n0 + wn * (TN * 16), N, tn * 2 + wn);
        }
    }
}

template <int TN, class EP>
int launch_conv_patch(const void* in, const void* Wt, int B, int H, int W, int C, int N, EP ep, hipStream_t s) {
    constexpr int PBN = 2 * TN * 16, P_LDS = 2 * P_BYTES + P_WST * PBN * BK * 2;
    const int tiles_x = (W + PT - 1) / PT, tiles_y = (H + PT - 1) / PT, tiles_n = (N + PBN - 1) / PBN;
    const long nwg = (long)B * tiles_x * tiles_y * tiles_n;
    if (nwg > 0x7fffffffL || (long)H * W * C > 0x7fffffffL || (long)PBN * 9 * C > 0x7fffffffL) return ISP_ERR_INVALID;
    static bool attr_done = false;
    auto kern = conv3x3_patch_kernel<EP, TN>;
    if (!attr_done) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, P_LDS) != hipSuccess)
            return ISP_ERR_LAUNCH;
        attr_done = true;
    }
    kern<<<(unsigned)nwg, 512, P_LDS, s>>>((const bf16_t*)in, (const bf16_t*)Wt, H, W, C, N, tiles_x, tiles_y, tiles_n,
                                          (int)nwg, ep);
    return isp_launch_status();
}

// One-wave-per-SIMD variant of the patch conv for 192-channel blocks: 4 waves, each 128 pixels x 96 channels
// (TM = 8, TN = 6: 14 fragment reads feed 48 MFMAs instead of 10 feeding 24, so the LDS port is ~60 % busy instead
// of ~100 %), up to 512 registers per wave: both K-halves' fragments are double-buffered and the loop is
// software-pipelined across the barrier -- the barrier of step s+1 sits inside the MFMA stream of step s (after the
// first 12 of its last 48 MFMAs), the fragments of step s+1 are read behind the remaining 36.  Weight ring of 3
// stages fetched two steps ahead (the DMA a barrier waits for was issued 1.5 steps earlier), counted vmcnt waits.
// Staging by buffer_load ... lds: 32-bit per-lane offsets against SGPR resources, out-of-image lanes use an
// out-of-range offset (the load returns zeros), so no 64-bit pointers or selects live in VGPRs.  The patch's 16-byte
// chunks are XOR-swizzled by the pixel's COLUMN in the patch ((px >> 1) & 7): a fragment read's address is then
// lane constant(dx, k-half) + immediate((t + dy + 1) * row pitch) -- no per-read address arithmetic.
//   LDS = 2 x 42 KiB patch + 3 x 24 KiB weights = 156 KiB.
template <class EP, int TN, bool F16 = false>  // F16: operands are IEEE half (same MFMA rate, 3 more mantissa bits)
__device__ __forceinline__ void conv3x3_patch4_body(const bf16_t* __restrict__ in, const bf16_t* __restrict__ Wt, int H,
                                                    int W, int C, int N, int tiles_x, int tiles_y, int tiles_n, int nwg,
                                                    const EP& ep) {
    constexpr int TM = 8, NWV = 4, PPW = (P_PIECES + NWV - 1) / NWV;          // 11 patch pieces per wave
    constexpr int PBN = 2 * TN * 16, PWB = PBN * BK * 2, WPW = PBN / 8 / NWV;  // 6 (TN = 6) or 4 weight pieces per wave
    constexpr int PPS = 2;                                                     // patch pieces per wave per K-step
    constexpr int ROWB = PW_ * 128;                                            // bytes per patch row
#ifndef ISP_C4_PRE
#define ISP_C4_PRE 2
#endif
#ifndef ISP_C4_SP
#define ISP_C4_SP 1
#endif
    constexpr int C4_PRE = ISP_C4_PRE, C4_SP = ISP_C4_SP;  // second-half MFMA rows before the barrier; MFMAs per fragment read
    static_assert(PPS * 6 >= PPW, "next patch must be complete before tap 6");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long K = 9L * C;
    const int cblocks = C / BK;
    const int per_img = tiles_x * tiles_y;
    // Persistent over output tiles: the grid is one workgroup per CU (ISEGPROBE_CONV_PERSIST=0: one per tile, as before)
    // and each walks tiles blockIdx, blockIdx + grid, ... -- a tile is ~60 us of MFMA work, and ending a workgroup +
    // starting the next (wave launch, resource setup, a cold first DMA round trip) cost several us of it.
#pragma unroll 1
    for (int wgi = blockIdx.x; wgi < nwg; wgi += gridDim.x) {
    const int wg = xcd_remap(wgi, nwg);
    const int tn = wg % tiles_n;
    const int tmi = wg / tiles_n;
    const int b = tmi / per_img, tt = tmi - b * per_img;
    const int y0 = (tt / tiles_x) * PT, x0 = (tt % tiles_x) * PT, n0 = tn * PBN;
    // Per-tile lane-dependent values (DMA offsets here, the epilogue's addresses below) are derived from an opaque copy
    // of the lane id: derived from `lane` itself their tile-invariant halves are hoisted out of the tile loop and, with
    // all 512 registers taken by the main loop, spilled to scratch there and reloaded for every tile (~200 values).
    int lane_t = lane;
    asm volatile("" : "+v"(lane_t));

    // --- DMA slots.  Wave w owns patch pieces w, w+4, ... (a wave whose last slot falls off the patch re-issues
    // its previous piece, so that every wave issues the same number of DMA instructions per step: the vmcnt
    // waits below count instructions); lane -> (pixel = 8*piece + lane/8, physical chunk lane%8).
    const __amdgpu_buffer_rsrc_t r_img =
        __builtin_amdgcn_make_buffer_rsrc((void*)(in + (size_t)b * H * W * C), 0, H * W * C * 2, 0x00020000);
    const int wrows = N - n0 < PBN ? N - n0 : PBN;
    const __amdgpu_buffer_rsrc_t r_w =
        __builtin_amdgcn_make_buffer_rsrc((void*)(Wt + (size_t)n0 * K), 0, (int)(wrows * K * 2), 0x00020000);
    unsigned poff[PPW];
    int ppiece[PPW];
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        const int piece = wid + NWV * i < P_PIECES ? wid + NWV * i : wid + NWV * (i - 1);
        const int pix = piece * 8 + (lane_t >> 3);
        const int py = pix / PW_, px = pix - py * PW_;
        const int iy = y0 - 1 + py, ix = x0 - 1 + px;
        const bool ok = pix < PPIX && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
        poff[i] = ok ? (unsigned)((iy * W + ix) * C + swz(px, lane_t & 7) * 8) * 2u : 0x80000000u;
        ppiece[i] = piece;
    }
    auto issue_patch = [&](int i, int cb, char* buf) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r_img, (ISP_LDS void*)(buf + ppiece[i] * 1024), 16, poff[i],
                                                 cb * (BK * 2), 0, 0);
    };
    unsigned woff[WPW];
#pragma unroll
    for (int i = 0; i < WPW; ++i) {
        const int row = (wid + NWV * i) * 8 + (lane_t >> 3);
        woff[i] = (unsigned)((row < wrows ? row : wrows - 1) * K + swz(row, lane_t & 7) * 8) * 2u;
    }
    auto issue_w = [&](int col, char* buf) {
#pragma unroll
        for (int i = 0; i < WPW; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r_w, (ISP_LDS void*)(buf + (wid + NWV * i) * 1024), 16, woff[i],
                                                     col * 2, 0, 0);
    };

    // --- fragment geometry
    const int wm = wid >> 1, wn = wid & 1;
    const int fr = lane & 15, fq = lane >> 4;
    int aoff[3][2];  // [dx + 1][k-half]: byte offset of patch pixel (row 8*wm, col fr + 1 + dx), swizzled chunk
#pragma unroll
    for (int d = 0; d < 3; ++d)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) aoff[d][ks] = (8 * wm * PW_ + fr + d) * 128 + swz(fr + d, ks * 4 + fq) * 16;
    const int w_off0 = (wn * (TN * 16) + fr) * 128 + swz(fr, fq) * 16;
    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    char* const s_w = smem + 2 * P_BYTES;
    auto read_a = [&](bf16x8 (&fa)[TM], const char* patch, int tap, int ks) {
        const int dy = tap / 3 - 1, dx = tap % 3 - 1;
        const char* base = patch + aoff[dx + 1][ks];
#pragma unroll
        for (int t = 0; t < TM; ++t) fa[t] = *reinterpret_cast<const bf16x8*>(base + (t + dy + 1) * ROWB);
    };
    auto read_w = [&](bf16x8 (&fw)[TN], const char* wb, int ks) {
#pragma unroll
        for (int t = 0; t < TN; ++t) fw[t] = *reinterpret_cast<const bf16x8*>(wb + ((w_off0 ^ (ks * 64)) + t * 2048));
    };
    auto mma = [&](const bf16x8 (&fa)[TM], const bf16x8 (&fw)[TN], int m_lo, int m_hi) {
#pragma unroll
        for (int mi = m_lo; mi < m_hi; ++mi)
#pragma unroll
            for (int ni = 0; ni < TN; ++ni)
                if constexpr (F16)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, fw[ni]),
                                                                        __builtin_bit_cast(f16x8_t, fa[mi]), acc[mi][ni], 0, 0, 0);
                else
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[ni], fa[mi], acc[mi][ni], 0, 0, 0);
    };

    // --- prologue: patch of channel block 0, weight tiles of steps 0 and 1
#pragma unroll
    for (int i = 0; i < PPW; ++i) issue_patch(i, 0, smem);
    issue_w(0, s_w);
    issue_w(C, s_w + PWB);  // step 1 = (cb 0, tap 1)
    bf16x8 fa0[TM], fa1[TM], fw0[TN], fw1[TN];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    read_w(fw0, s_w, 0);
    read_a(fa0, smem, 0, 0);

    // One K-step (tap TAP of channel block cb); ring slot of the step = TAP % 3 (9 % 3 == 0).
    auto step = [&]<int TAP, bool MORE, bool HASW>(int cb, const char* patch, char* patch_next) {
        constexpr int NP = !MORE ? 0 : (PPS * TAP + PPS <= PPW ? PPS : (PPS * TAP < PPW ? PPW - PPS * TAP : 0));
        // The instruction order is pinned (sched_barrier / sched_group_barrier): with one wave per SIMD nothing
        // else covers a fragment read that is issued right before its first use.
        // region 1: DMA for two steps ahead + second-half fragment reads, one per MFMA, inside the 48 first-half MFMAs
        constexpr int NDMA = NP + (HASW ? WPW : 0);
#ifdef ISP_ABLATE_NO_DMA  // timing experiment only
        constexpr int NDMA_ISSUED = 0;
#else
        constexpr int NDMA_ISSUED = NDMA;
#pragma unroll
        for (int i = PPS * TAP; i < PPS * TAP + NP; ++i) issue_patch(i, cb + 1, patch_next);
        if constexpr (HASW)
            issue_w(TAP < 7 ? (TAP + 2) * C + cb * BK : (TAP - 7) * C + (cb + 1) * BK, s_w + ((TAP + 2) % 3) * PWB);
#endif
        read_w(fw1, s_w + (TAP % 3) * PWB, 1);
        read_a(fa1, patch, TAP, 1);
        mma(fa0, fw0, 0, TM);
#pragma unroll
        for (int i = 0; i < NDMA_ISSUED; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // 1 MFMA
            __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);  // 1 VMEM (LDS-DMA)
        }
#pragma unroll
        for (int i = 0; i < TM + TN; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, C4_SP, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // 1 DS read
        }
        __builtin_amdgcn_sched_group_barrier(0x008, TM * TN - NDMA_ISSUED - C4_SP * (TM + TN), 0);
        __builtin_amdgcn_sched_barrier(0);
        // region 2: the first 12 second-half MFMAs, then the barrier of step s+1: everything older than this step's
        // DMA has landed (weight tile of step s+1, earlier patch pieces) and every wave is done with tile s-1
        mma(fa1, fw1, 0, C4_PRE);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA_ISSUED) : "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        // region 3: first-half fragments of step s+1 behind the remaining 36 MFMAs
        // (after the very last step these read stale, in-bounds LDS: unused)
        read_w(fw0, s_w + ((TAP + 1) % 3) * PWB, 0);
        read_a(fa0, TAP < 8 ? patch : patch_next, TAP < 8 ? TAP + 1 : 0, 0);
        mma(fa1, fw1, C4_PRE, TM);
#pragma unroll
        for (int i = 0; i < TM + TN; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, C4_SP, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, (TM - C4_PRE) * TN - C4_SP * (TM + TN), 0);
        __builtin_amdgcn_sched_barrier(0);
    };
    auto block9 = [&]<bool MORE>(int cb) {
        const char* patch = smem + (cb & 1) * P_BYTES;
        char* patch_next = smem + ((cb + 1) & 1) * P_BYTES;
        step.template operator()<0, MORE, true>(cb, patch, patch_next);
        step.template operator()<1, MORE, true>(cb, patch, patch_next);
        step.template operator()<2, MORE, true>(cb, patch, patch_next);
        step.template operator()<3, MORE, true>(cb, patch, patch_next);
        step.template operator()<4, MORE, true>(cb, patch, patch_next);
        step.template operator()<5, MORE, true>(cb, patch, patch_next);
        step.template operator()<6, MORE, true>(cb, patch, patch_next);
        step.template operator()<7, MORE, MORE>(cb, patch, patch_next);  // steps s+2 of taps 7, 8 are in block cb+1
        step.template operator()<8, MORE, MORE>(cb, patch, patch_next);
    };
    for (int cb = 0; cb + 1 < cblocks; ++cb) block9.template operator()<true>(cb);
    block9.template operator()<false>(cblocks - 1);

    auto row_of = [&](int r) -> long {
        const int y = y0 + (r >> 4), x = x0 + (r & 15);
        return (y < H && x < W) ? ((long)b * H + y) * W + x : -1;
    };
    int lane_e = lane;  // (opaque, as lane_t above)
    asm volatile("" : "+v"(lane_e));
    const int fr_e = lane_e & 15, fq_e = lane_e >> 4;
#ifdef ISP_ABLATE_NO_EPILOGUE  // timing experiment only: the persistent loop without its epilogues
    if (acc[0][0][0] != 12345.678f) {
    } else
#endif
    if (y0 + PT <= H && x0 + PT <= W && n0 + PBN <= N) {  // interior tile: no per-lane checks
        auto row_in = [&](int r) -> long { return ((long)b * H + y0 + (r >> 4)) * W + x0 + (r & 15); };
        // the other waves are past every LDS read whose value is used: no barrier before the wave-private staging
        if constexpr (kStagedStore<EP>) {
            staged_epilogue<TM, TN>(ep, acc, row_in, wm * (TM * 16), fr_e, fq_e, lane_e, n0 + wn * (TN * 16),
                                    smem + wid * kStageBytes<TM, TN>, tn * 2 + wn);
        } else {
            run_epilogue<TM, TN, true>(ep, acc, row_in, wm * (TM * 16), fr_e, fq_e, n0 + wn * (TN * 16), N, tn * 2 + wn);
        }
    } else {
        run_epilogue<TM, TN>(ep, acc, row_of, wm * (TM * 16), fr_e, fq_e, n0 + wn * (TN * 16), N, tn * 2 + wn);
    }
    if (wgi + (int)gridDim.x < nwg) {
        // the next tile's prologue DMA overwrites LDS that the waves' epilogue staging may still be reading
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    }  // tiles
}

// (thin kernels: with the body above written directly as a __global__ template that depends on TN, hipcc 7.2
// silently leaves the kernel's host-side handle undefined)
template <class EP, bool F16 = false>
__global__ __launch_bounds__(256, 1) void conv3x3_patch4_kernel_192(const bf16_t* __restrict__ in,
                                                                    const bf16_t* __restrict__ Wt, int H, int W, int C,
                                                                    int N, int tiles_x, int tiles_y, int tiles_n,
                                                                    int nwg, EP ep) {
    conv3x3_patch4_body<EP, 6, F16>(in, Wt, H, W, C, N, tiles_x, tiles_y, tiles_n, nwg, ep);
}
template <class EP, bool F16 = false>
__global__ __launch_bounds__(256, 1) void conv3x3_patch4_kernel_128(const bf16_t* __restrict__ in,
                                                                    const bf16_t* __restrict__ Wt, int H, int W, int C,
                                                                    int N, int tiles_x, int tiles_y, int tiles_n,
                                                                    int nwg, EP ep) {
    conv3x3_patch4_body<EP, 4, F16>(in, Wt, H, W, C, N, tiles_x, tiles_y, tiles_n, nwg, ep);
}

template <int TN, class EP, bool F16 = false>
int launch_conv_patch4(const void* in, const void* Wt, int B, int H, int W, int C, int N, EP ep, hipStream_t s) {
    constexpr int PBN = 2 * TN * 16, P_LDS = 2 * P_BYTES + 3 * PBN * BK * 2;
    const int tiles_x = (W + PT - 1) / PT, tiles_y = (H + PT - 1) / PT, tiles_n = (N + PBN - 1) / PBN;
    const long nwg = (long)B * tiles_x * tiles_y * tiles_n;
    // buffer resources address bytes with 32-bit offsets; 0x80000000 must stay out of range
    if (nwg > 0x7fffffffL || (long)H * W * C * 2 >= 0x7fffffffL || (long)PBN * 9 * C * 2 >= 0x7fffffffL) return ISP_ERR_UNSUPPORTED;
    static bool attr_done = false;
    static_assert(TN == 6 || TN == 4);
    void (*kern)(const bf16_t*, const bf16_t*, int, int, int, int, int, int, int, int, EP);
    if constexpr (TN == 6) kern = conv3x3_patch4_kernel_192<EP, F16>;
    else kern = conv3x3_patch4_kernel_128<EP, F16>;
    if (!attr_done) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, P_LDS) != hipSuccess)
            return ISP_ERR_LAUNCH;
        attr_done = true;
    }
    static int persist_grid = -1;  // workgroups of the persistent launch = CUs (156 KiB of LDS: one workgroup per CU)
    if (persist_grid < 0) {
        const char* e = getenv("ISEGPROBE_CONV_PERSIST");
        int dev = 0;
        hipDeviceProp_t p;
        if (e && e[0] == '0') persist_grid = 0;
        else if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) persist_grid = p.multiProcessorCount;
        else persist_grid = 0;
    }
    const unsigned grid = persist_grid > 0 && nwg > persist_grid ? (unsigned)persist_grid : (unsigned)nwg;
    kern<<<grid, 256, P_LDS, s>>>((const bf16_t*)in, (const bf16_t*)Wt, H, W, C, N, tiles_x, tiles_y, tiles_n, (int)nwg, ep);
    return isp_launch_status();
}

static int conv_patch_waves() {  // A/B switch: ISEGPROBE_CONV_ENGINE=8 keeps 192-channel blocks on the 8-wave patch kernel
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("ISEGPROBE_CONV_ENGINE");
        v = (e && e[0] == '8') ? 8 : 4;
    }
    return v;
}

// dispatch of the epilogue kinds the patch conv supports
template <int TN>
int dispatch_conv_patch(const void* in, const void* Wt, int B, int H, int W, int C, int N, const isp_epilogue* e,
                        hipStream_t s) {
    const long M = (long)B * H * W;
    const long ldo = e->ldo > 0 ? e->ldo : N;
    {
        // one-wave-per-SIMD kernel (it stores its bf16 tile as 16-byte chunks)
        if (conv_patch_waves() == 4 && ldo % 8 == 0 && (reinterpret_cast<size_t>(e->out) & 15) == 0) {
            switch (e->kind) {
                case ISP_EP_BIAS_BF16:
                    return launch_conv_patch4<TN>(in, Wt, B, H, W, C, N, EpBiasActBf16<ACT_NONE>{(bf16_t*)e->out, e->bias, ldo}, s);
                case ISP_EP_BIAS_RELU_BF16:
                    return launch_conv_patch4<TN>(in, Wt, B, H, W, C, N, EpBiasActBf16<ACT_RELU>{(bf16_t*)e->out, e->bias, ldo}, s);
                case ISP_EP_BIAS_TAPS_RELU_BF16:
                    if (!e->bias || !e->pos || e->img_h <= 0 || e->img_w <= 0 || ldo != N) return ISP_ERR_INVALID;
                    return launch_conv_patch4<TN>(in, Wt, B, H, W, C, N,
                                              EpBiasTapsReluBf16<>{(bf16_t*)e->out, e->bias, e->pos, e->img_h, e->img_w, ldo}, s);
                case ISP_EP_RELU_DOT_PARTIAL_F32:
                    if (!e->bias || !e->gamma) return ISP_ERR_INVALID;
                    return launch_conv_patch4<TN>(in, Wt, B, H, W, C, N, EpReluDotPartial{(float*)e->out, e->bias, e->gamma, M}, s);
                default: break;
            }
        }
    }
    switch (e->kind) {
        case ISP_EP_BIAS_BF16:
            return launch_conv_patch<TN>(in, Wt, B, H, W, C, N, EpBiasActBf16<ACT_NONE>{(bf16_t*)e->out, e->bias, ldo}, s);
        case ISP_EP_BIAS_RELU_BF16:
            return launch_conv_patch<TN>(in, Wt, B, H, W, C, N, EpBiasActBf16<ACT_RELU>{(bf16_t*)e->out, e->bias, ldo}, s);
        case ISP_EP_BIAS_TAPS_RELU_BF16:
            if (!e->bias || !e->pos || e->img_h <= 0 || e->img_w <= 0 || ldo != N) return ISP_ERR_INVALID;
            return launch_conv_patch<TN>(in, Wt, B, H, W, C, N,
                                     EpBiasTapsReluBf16<>{(bf16_t*)e->out, e->bias, e->pos, e->img_h, e->img_w, ldo}, s);
        case ISP_EP_RELU_DOT_PARTIAL_F32:
            if (!e->bias || !e->gamma) return ISP_ERR_INVALID;
            return launch_conv_patch<TN>(in, Wt, B, H, W, C, N, EpReluDotPartial{(float*)e->out, e->bias, e->gamma, M}, s);
        default:
            return ISP_ERR_UNSUPPORTED;
    }
}

template <class CFG, class AL, unsigned KINDS = 0xffffffffu>  // KINDS: bit mask of epilogue kinds to instantiate
int dispatch_epilogue(AL al, const void* Wt, long M, int N, int K, const isp_epilogue* e, hipStream_t s) {
    if (!e || !e->out) return ISP_ERR_INVALID;
    if (e->kind < 0 || e->kind > 31 || !((KINDS >> e->kind) & 1u)) return ISP_ERR_UNSUPPORTED;
    const long ldo = e->ldo > 0 ? e->ldo : N;
    if (ldo % 4 != 0) return ISP_ERR_INVALID;
    switch (e->kind) {
        case ISP_EP_BIAS_BF16:
            if constexpr (!((KINDS >> ISP_EP_BIAS_BF16) & 1u)) return ISP_ERR_UNSUPPORTED; else {
            return launch_gemm<CFG>(al, Wt, M, N, K, EpBiasActBf16<ACT_NONE>{(bf16_t*)e->out, e->bias, ldo}, s);
            }
        case ISP_EP_BIAS_RELU_BF16:
            if constexpr (!((KINDS >> ISP_EP_BIAS_RELU_BF16) & 1u)) return ISP_ERR_UNSUPPORTED; else {
            return launch_gemm<CFG>(al, Wt, M, N, K, EpBiasActBf16<ACT_RELU>{(bf16_t*)e->out, e->bias, ldo}, s);
            }
        case ISP_EP_BIAS_GELU_BF16:
            if constexpr (!((KINDS >> ISP_EP_BIAS_GELU_BF16) & 1u)) return ISP_ERR_UNSUPPORTED; else {
            return launch_gemm<CFG>(al, Wt, M, N, K, EpBiasActBf16<ACT_GELU>{(bf16_t*)e->out, e->bias, ldo}, s);
            }
        case ISP_EP_BIAS_F32:
            if constexpr (!((KINDS >> ISP_EP_BIAS_F32) & 1u)) return ISP_ERR_UNSUPPORTED; else {
            return launch_gemm<CFG>(al, Wt, M, N, K, EpBiasActF32<ACT_NONE>{(float*)e->out, e->bias, ldo}, s);
            }
        case ISP_EP_RESIDUAL_F32:
            if constexpr (!((KINDS >> ISP_EP_RESIDUAL_F32) & 1u)) return ISP_ERR_UNSUPPORTED; else {
            return launch_gemm<CFG>(al, Wt, M, N, K, EpResidual{(float*)e->out, e->bias, e->gamma, ldo}, s);
            }
        case ISP_EP_TOKENS_F32:
            if constexpr (!((KINDS >> ISP_EP_TOKENS_F32) & 1u)) return ISP_ERR_UNSUPPORTED; else {
            if (e->tokens_per_image <= 0 || !e->bias) return ISP_ERR_INVALID;
            return launch_gemm<CFG>(al, Wt, M, N, K, EpTokens{(float*)e->out, e->bias, e->pos, e->tokens_per_image, ldo}, s);
            }
        case ISP_EP_AXPY_RES_BF16:
            if constexpr (!((KINDS >> ISP_EP_AXPY_RES_BF16) & 1u)) return ISP_ERR_UNSUPPORTED; else {
            if (!e->res) return ISP_ERR_INVALID;
            return launch_gemm<CFG>(al, Wt, M, N, K,
                               EpAxpyResBf16<>{(bf16_t*)e->out, (const bf16_t*)e->res, e->bias, e->alpha, ldo}, s);
            }
        case ISP_EP_BIAS_TAPS_RELU_BF16:
            if constexpr (!((KINDS >> ISP_EP_BIAS_TAPS_RELU_BF16) & 1u)) return ISP_ERR_UNSUPPORTED; else {
            if (!e->bias || !e->pos || e->img_h <= 0 || e->img_w <= 0 || ldo != N) return ISP_ERR_INVALID;
            return launch_gemm<CFG>(al, Wt, M, N, K,
                               EpBiasTapsReluBf16<>{(bf16_t*)e->out, e->bias, e->pos, e->img_h, e->img_w, ldo}, s);
            }
        case ISP_EP_BIAS_QGELU_BF16:
            if constexpr (!((KINDS >> ISP_EP_BIAS_QGELU_BF16) & 1u)) return ISP_ERR_UNSUPPORTED; else {
            return launch_gemm<CFG>(al, Wt, M, N, K, EpBiasActBf16<ACT_QGELU>{(bf16_t*)e->out, e->bias, ldo}, s);
            }
        case ISP_EP_RELU_DOT_PARTIAL_F32:
            if constexpr (!((KINDS >> ISP_EP_RELU_DOT_PARTIAL_F32) & 1u)) return ISP_ERR_UNSUPPORTED; else {
            if (!e->bias || !e->gamma) return ISP_ERR_INVALID;
            return launch_gemm<CFG>(al, Wt, M, N, K, EpReluDotPartial{(float*)e->out, e->bias, e->gamma, M}, s);
            }
        case ISP_EP_BIAS_GELU_SAVE_BF16:
            if constexpr (!((KINDS >> ISP_EP_BIAS_GELU_SAVE_BF16) & 1u)) return ISP_ERR_UNSUPPORTED; else {
            if (!e->out2) return ISP_ERR_INVALID;
            return launch_gemm<CFG>(al, Wt, M, N, K, EpBiasGeluSaveBf16<ACT_GELU>{(bf16_t*)e->out, (bf16_t*)e->out2, e->bias, ldo}, s);
            }
        case ISP_EP_BIAS_QGELU_SAVE_BF16:
            if constexpr (!((KINDS >> ISP_EP_BIAS_QGELU_SAVE_BF16) & 1u)) return ISP_ERR_UNSUPPORTED; else {
            if (!e->out2) return ISP_ERR_INVALID;
            return launch_gemm<CFG>(al, Wt, M, N, K, EpBiasGeluSaveBf16<ACT_QGELU>{(bf16_t*)e->out, (bf16_t*)e->out2, e->bias, ldo}, s);
            }
        case ISP_EP_MUL_DGELU_BF16:
            if constexpr (!((KINDS >> ISP_EP_MUL_DGELU_BF16) & 1u)) return ISP_ERR_UNSUPPORTED; else {
            if (!e->res) return ISP_ERR_INVALID;
            return launch_gemm<CFG>(al, Wt, M, N, K, EpMulDGeluBf16<ACT_GELU>{(bf16_t*)e->out, (const bf16_t*)e->res, ldo}, s);
            }
        case ISP_EP_MUL_DQGELU_BF16:
            if constexpr (!((KINDS >> ISP_EP_MUL_DQGELU_BF16) & 1u)) return ISP_ERR_UNSUPPORTED; else {
            if (!e->res) return ISP_ERR_INVALID;
            return launch_gemm<CFG>(al, Wt, M, N, K, EpMulDGeluBf16<ACT_QGELU>{(bf16_t*)e->out, (const bf16_t*)e->res, ldo}, s);
            }
        default:
            return ISP_ERR_UNSUPPORTED;
    }
}

}  // namespace

extern "C" int isp_gemm_bf16(const void* A, long lda, const void* Wt, long M, int N, int K, const isp_epilogue* ep,
                             void* stream) {
    ISP_CHECK_ARG(A && Wt && lda >= K && lda % 8 == 0);
    auto run = [&](auto cfg) {
        using CFG = decltype(cfg);
        DenseA<CFG::PA> al;
        al.A = (const bf16_t*)A;
        al.lda = lda;
        al.M = M;
        return dispatch_epilogue<CFG, DenseA<CFG::PA>, 0x3e7fu>(al, Wt, M, N, K, ep, (hipStream_t)stream);
    };
    // Short-K GEMMs (6-8 K-steps) are bound by per-tile latencies (first-stage DMA, epilogue chain), not by MFMA or
    // staging bandwidth: without their epilogue they run at 800-980 TFLOP/s, with it at 520-650 (M = 1.6 M rows,
    // 256x192 tile) / 680-790 (ViT-sized, 128x128, 2 blocks per CU).  Measured alternatives on the M = 1.6 M shapes:
    // 256x128 with a 3-stage ring 523-619 (ring 2: 451-535), 128x128 ring 3 (1 block / CU) 448-505, 128x64 and
    // 64x128 ring 3 (2 blocks / CU) 405-522; ViT-sized problems lose 10 % on the 8-wave tiles and stay on 128x128.
    if ((M + 255) / 256 >= 512 && N >= 384) return run(CfgConv192{});
    // the batch-2 click loop (M = 2050 tokens): 128-row tiles give 153 blocks for 256 CUs; 64-row tiles fill the chip
    if (((M + 127) / 128) * ((N + 127) / 128) < 256) return run(Cfg64{});
    return run(Cfg128{});
}

// IEEE-half operands and 16-bit outputs (LoftUp's inference stream: three more mantissa bits on every operand and map of
// its two cross-attention + feed-forward layers).  Epilogues: bias, bias + GELU, residual add (res / out half).
extern "C" int isp_gemm_f16(const void* A, long lda, const void* Wt, long M, int N, int K, const isp_epilogue* e, void* stream) {
    ISP_CHECK_ARG(A && Wt && e && e->out && lda >= K && lda % 8 == 0);
    const long ldo = e->ldo > 0 ? e->ldo : N;
    if (ldo % 4 != 0) return ISP_ERR_INVALID;
    hipStream_t s = (hipStream_t)stream;
    auto run = [&](auto cfg) {
        using CFG = decltype(cfg);
        DenseA<CFG::PA> al;
        al.A = (const bf16_t*)A;
        al.lda = lda;
        al.M = M;
        switch (e->kind) {
            case ISP_EP_BIAS_BF16:
                return launch_gemm<CFG, true>(al, Wt, M, N, K, EpBiasActBf16<ACT_NONE, true>{(bf16_t*)e->out, e->bias, ldo}, s);
            case ISP_EP_BIAS_GELU_BF16:
                return launch_gemm<CFG, true>(al, Wt, M, N, K, EpBiasActBf16<ACT_GELU, true>{(bf16_t*)e->out, e->bias, ldo}, s);
            case ISP_EP_AXPY_RES_BF16:
                if (!e->res) return (int)ISP_ERR_INVALID;
                return launch_gemm<CFG, true>(al, Wt, M, N, K,
                                              EpAxpyResBf16<true>{(bf16_t*)e->out, (const bf16_t*)e->res, e->bias, e->alpha, ldo}, s);
            case ISP_EP_RESIDUAL_F32:  // the ViT's fp32 residual stream: x += gamma * (A W^T + bias) with half operands
                return launch_gemm<CFG, true>(al, Wt, M, N, K, EpResidual{(float*)e->out, e->bias, e->gamma, ldo}, s);
            case ISP_EP_RESIDUAL_STATS_F32:
                if (!e->out2 || !e->out3) return (int)ISP_ERR_INVALID;
                return launch_gemm<CFG, true>(al, Wt, M, N, K, EpResidualStats{(float*)e->out, e->bias, e->gamma, ldo, (unsigned short*)e->out3, (float*)e->out2, M}, s);
            case ISP_EP_LNFOLD_BF16:
            case ISP_EP_LNFOLD_GELU_BF16: {
                if (!e->bias || !e->gamma || !e->res || e->tokens_per_image <= 0 || e->img_h <= 0 || e->img_h > 8) return (int)ISP_ERR_INVALID;  // (<= 8 statistics slots)
                const float inv_d = 1.0f / (float)e->tokens_per_image;
                if (e->kind == ISP_EP_LNFOLD_BF16)
                    return launch_gemm<CFG, true>(al, Wt, M, N, K,
                                                  EpLnFold<ACT_NONE, true>{(bf16_t*)e->out, e->bias, e->gamma, (const float*)e->res, M, e->img_h, inv_d, e->alpha, ldo}, s);
                return launch_gemm<CFG, true>(al, Wt, M, N, K,
                                              EpLnFold<ACT_GELU, true>{(bf16_t*)e->out, e->bias, e->gamma, (const float*)e->res, M, e->img_h, inv_d, e->alpha, ldo}, s);
            }
            default:
                return (int)ISP_ERR_UNSUPPORTED;
        }
    };
    if (e->kind == ISP_EP_LNFOLD_LAYERNORM_BF16) {  // one tile spans the output row: smallest configuration with N <= BN
        if (!e->bias || !e->gamma || !e->res || !e->pos || !e->out2 || e->tokens_per_image <= 0 || e->img_h <= 0 || e->img_h > 8 || N > CfgWide512::BN)
            return ISP_ERR_INVALID;
        auto go = [&](auto cfg) {
            using CFG = decltype(cfg);
            static_assert(CFG::WN * CFG::BM * 8 <= CFG::LDS);
            DenseA<CFG::PA> al;
            al.A = (const bf16_t*)A;
            al.lda = lda;
            al.M = M;
            return launch_gemm<CFG, true>(al, Wt, M, N, K,
                                          EpLnFoldLayerNorm<true>{(bf16_t*)e->out, e->bias, e->gamma, (const float*)e->res, M, e->img_h,
                                                                  1.0f / (float)e->tokens_per_image, e->alpha, ldo, e->pos,
                                                                  (const float*)e->out2, 1.0f / (float)N, e->alpha2}, s);
        };
        if (N <= 128) return go(Cfg128{});
        if (N <= 384) return go(CfgWide384{});
        if (N <= 448) return go(CfgWide448{});
        return go(CfgWide512{});
    }
    if (e->kind == ISP_EP_AXPY_RES_STATS_BF16) {  // full-row tiles whatever M is: the statistics' slot count is then fixed
        if (!e->res || !e->out2 || N > CfgWide448::BN) return ISP_ERR_INVALID;
        DenseA<CfgWide448::PA> al;
        al.A = (const bf16_t*)A;
        al.lda = lda;
        al.M = M;
        return launch_gemm<CfgWide448, true>(al, Wt, M, N, K,
                                             EpAxpyResStats<true>{(bf16_t*)e->out, (const bf16_t*)e->res, e->bias, e->alpha, ldo, (float*)e->out2, M}, s);
    }
    static const bool wide_off = [] { const char* e = getenv("ISEGPROBE_GEMM_WIDE"); return e && e[0] == '0'; }();
    if ((M + 127) / 128 >= 1024 && !wide_off) {  // (row tiles fill the chip four times over)
        if (N > 256 && N <= 384) return run(CfgWide384{});
        if (N > 384 && N <= 448) return run(CfgWide448{});
        if (N > 448 && N <= 512) return run(CfgWide512{});
    }
    if ((M + 255) / 256 >= 512 && N >= 384) return run(CfgConv192{});
    if (((M + 127) / 128) * ((N + 127) / 128) < 256) return run(Cfg64{});
    return run(Cfg128{});
}

// partial-statistics slots isp_gemm_f16 writes with ISP_EP_AXPY_RES_STATS_BF16 ([slots][M][2] floats)
extern "C" int isp_gemm_stats_slots(void) { return CfgWide448::WN; }
// ... and with ISP_EP_RESIDUAL_STATS_F32, where the tile configuration follows the problem size (same rules as isp_gemm_f16)
template <class CFG>
static int stats_slots_of(int N) { return ((N + CFG::BN - 1) / CFG::BN) * CFG::WN; }
extern "C" int isp_gemm_f16_stats_slots(long M, int N) {
    static const bool wide_off = [] { const char* e = getenv("ISEGPROBE_GEMM_WIDE"); return e && e[0] == '0'; }();
    if ((M + 127) / 128 >= 1024 && !wide_off) {
        if (N > 256 && N <= 384) return stats_slots_of<CfgWide384>(N);
        if (N > 384 && N <= 448) return stats_slots_of<CfgWide448>(N);
        if (N > 448 && N <= 512) return stats_slots_of<CfgWide512>(N);
    }
    if ((M + 255) / 256 >= 512 && N >= 384) return stats_slots_of<CfgConv192>(N);
    if (((M + 127) / 128) * ((N + 127) / 128) < 256) return stats_slots_of<Cfg64>(N);
    return stats_slots_of<Cfg128>(N);
}

// A/B switch for experiments: ISEGPROBE_CONV_ENGINE=tile selects the generic tile engine for every conv
static bool ep_forces_tile_engine() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("ISEGPROBE_CONV_ENGINE");
        v = (e && e[0] == 't') ? 1 : 0;
    }
    return v == 1;
}

// 128-channel blocks of the patch kernel: exact tilings, or a ragged last block that wastes <= 15 % (LoftUp's 448)
static bool patch128_ok(int N) { return N >= 128 && ((N + 127) / 128 * 128 - N) * 100 <= 15 * N; }

// number of partial-sum slots isp_conv3x3_nhwc_bf16 writes with ISP_EP_RELU_DOT_PARTIAL_F32
extern "C" int isp_conv3x3_partial_slots(int N) {
    if (N % 192 == 0) return ((N + CfgConv192::BN - 1) / CfgConv192::BN) * CfgConv192::WN;
    if (patch128_ok(N) && !ep_forces_tile_engine()) return (N + 127) / 128 * 2;  // patch kernel, 128-channel blocks
    if (N > 64) return ((N + CfgConv128::BN - 1) / CfgConv128::BN) * CfgConv128::WN;
    return ((N + Cfg128::BN - 1) / Cfg128::BN) * Cfg128::WN;
}

// IEEE-half operands (in, Wt) and 16-bit outputs: the head's convolutions behind the FeatUp-JBU stack, whose maps are
// half already -- three more mantissa bits than bf16 on the head's inputs, weights and hidden map (the largest
// contribution to the bf16 path's logit error, tools/diag_precision_full.py) at the same MFMA rate.  192-channel-block
// patch kernel only: N % 192 == 0, C % 64 == 0, 16-byte aligned output with ldo % 8 == 0.
extern "C" int isp_conv3x3_nhwc_f16(const void* in, const void* Wt, int B, int H, int W, int C, int N, const isp_epilogue* e,
                                    void* stream) {
    ISP_CHECK_ARG(in && Wt && e && e->out && B > 0 && H > 0 && W > 0 && C > 0 && C % BK == 0 && N > 0);
    const long M = (long)B * H * W;
    ISP_CHECK_ARG(M <= 0x7fffffffL);
    const long ldo = e->ldo > 0 ? e->ldo : N;
    if (ldo % 8 != 0 || (reinterpret_cast<size_t>(e->out) & 15) != 0) return ISP_ERR_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    if (N % 192 != 0) {  // 128-channel blocks (LoftUp's 448-channel maps, 1024- and 128-wide heads)
        if (!patch128_ok(N)) return ISP_ERR_UNSUPPORTED;
        switch (e->kind) {
            case ISP_EP_BIAS_BF16:
                return launch_conv_patch4<4, EpBiasActBf16<ACT_NONE, true>, true>(in, Wt, B, H, W, C, N, {(bf16_t*)e->out, e->bias, ldo}, s);
            case ISP_EP_BIAS_RELU_BF16:
                return launch_conv_patch4<4, EpBiasActBf16<ACT_RELU, true>, true>(in, Wt, B, H, W, C, N, {(bf16_t*)e->out, e->bias, ldo}, s);
            case ISP_EP_BIAS_TAPS_RELU_BF16:  // (heads of widths that tile into 128-channel blocks: ViT-L's 1024, the 128-wide fixtures)
                if (!e->bias || !e->pos || e->img_h <= 0 || e->img_w <= 0 || ldo != N) return ISP_ERR_INVALID;
                return launch_conv_patch4<4, EpBiasTapsReluBf16<true>, true>(in, Wt, B, H, W, C, N,
                                                                             {(bf16_t*)e->out, e->bias, e->pos, e->img_h, e->img_w, ldo}, s);
            case ISP_EP_RELU_DOT_PARTIAL_F32:
                if (!e->bias || !e->gamma) return ISP_ERR_INVALID;
                return launch_conv_patch4<4, EpReluDotPartial, true>(in, Wt, B, H, W, C, N, {(float*)e->out, e->bias, e->gamma, M}, s);
            case ISP_EP_BIAS_RELU_STATS_BF16:
                if (!e->out2) return ISP_ERR_INVALID;
                return launch_conv_patch4<4, EpBiasActStats<ACT_RELU, true>, true>(in, Wt, B, H, W, C, N,
                                                                                   {{(bf16_t*)e->out, e->bias, ldo}, (float*)e->out2, M}, s);
            default:
                return ISP_ERR_UNSUPPORTED;
        }
    }
    switch (e->kind) {
        case ISP_EP_BIAS_RELU_STATS_BF16:
            if (!e->out2) return ISP_ERR_INVALID;
            return launch_conv_patch4<6, EpBiasActStats<ACT_RELU, true>, true>(in, Wt, B, H, W, C, N,
                                                                               {{(bf16_t*)e->out, e->bias, ldo}, (float*)e->out2, M}, s);
        case ISP_EP_BIAS_BF16:
            return launch_conv_patch4<6, EpBiasActBf16<ACT_NONE, true>, true>(in, Wt, B, H, W, C, N, {(bf16_t*)e->out, e->bias, ldo}, s);
        case ISP_EP_BIAS_RELU_BF16:
            return launch_conv_patch4<6, EpBiasActBf16<ACT_RELU, true>, true>(in, Wt, B, H, W, C, N, {(bf16_t*)e->out, e->bias, ldo}, s);
        case ISP_EP_BIAS_TAPS_RELU_BF16:
            if (!e->bias || !e->pos || e->img_h <= 0 || e->img_w <= 0 || ldo != N) return ISP_ERR_INVALID;
            return launch_conv_patch4<6, EpBiasTapsReluBf16<true>, true>(in, Wt, B, H, W, C, N,
                                                                         {(bf16_t*)e->out, e->bias, e->pos, e->img_h, e->img_w, ldo}, s);
        case ISP_EP_RELU_DOT_PARTIAL_F32:
            if (!e->bias || !e->gamma) return ISP_ERR_INVALID;
            return launch_conv_patch4<6, EpReluDotPartial, true>(in, Wt, B, H, W, C, N, {(float*)e->out, e->bias, e->gamma, M}, s);
        default:
            return ISP_ERR_UNSUPPORTED;
    }
}

// partial-statistics slots isp_conv3x3_nhwc_f16 writes with ISP_EP_BIAS_RELU_STATS_BF16 ([slots][B*H*W][2] floats)
extern "C" int isp_conv_stats_slots(int N) { return N % 192 == 0 ? (N / 192) * 2 : ((N + 127) / 128) * 2; }

extern "C" int isp_conv3x3_nhwc_bf16(const void* in, const void* Wt, int B, int H, int W, int C, int N,
                                     const isp_epilogue* ep, void* stream) {
    ISP_CHECK_ARG(in && Wt && B > 0 && H > 0 && W > 0 && C > 0 && C % BK == 0);
    const long M = (long)B * H * W;
    ISP_CHECK_ARG(M <= 0x7fffffffL);
    constexpr unsigned CONV_KINDS = (1u << ISP_EP_BIAS_BF16) | (1u << ISP_EP_BIAS_RELU_BF16) | (1u << ISP_EP_BIAS_F32) |
                                    (1u << ISP_EP_BIAS_TAPS_RELU_BF16) | (1u << ISP_EP_RELU_DOT_PARTIAL_F32);
    auto run = [&](auto cfg) {
        using CFG = decltype(cfg);
        Conv3x3A<CFG::PA> al;
        al.in = (const bf16_t*)in;
        al.H = H;
        al.W = W;
        al.C = C;
        al.M = M;
        al.cblocks = C / BK;
        al.tiles_x = (W + 15) / 16;
        al.tiles_y = (H + CFG::BM / 16 - 1) / (CFG::BM / 16);
        return dispatch_epilogue<CFG, Conv3x3A<CFG::PA>, CONV_KINDS>(al, Wt, M, N, 9 * C, ep, (hipStream_t)stream);
    };
    // LDS-resident-patch kernel when 192- or 128-channel blocks tile N exactly (C = N = 384 / 768 heads: 192;
    // N = 1024, ViT-L heads: 128.  A 256-channel variant, TN = 8, needs ~280 VGPRs and spills.)
    if (!ep_forces_tile_engine() && (N % 192 == 0 || patch128_ok(N))) {
        if (!ep) return ISP_ERR_INVALID;
        const int rc = N % 192 == 0 ? dispatch_conv_patch<6>(in, Wt, B, H, W, C, N, ep, (hipStream_t)stream)
                                    : dispatch_conv_patch<4>(in, Wt, B, H, W, C, N, ep, (hipStream_t)stream);
        if (rc != ISP_ERR_UNSUPPORTED) return rc;
    }
    if (N % 192 == 0) return run(CfgConv192{});
    if (N > 64) return run(CfgConv128{});
    return run(Cfg128{});
}
