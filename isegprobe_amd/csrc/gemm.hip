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

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = BM * BK * 2;        // 16 KiB per operand tile
constexpr int STAGE_BYTES = 2 * TILE_BYTES;    // A + W
constexpr int LDS_BYTES = 2 * STAGE_BYTES;     // double buffered: 64 KiB -> 2 blocks / CU

// ------------------------------------------------------------------------------ A loaders
// A loader contract: init(row_slot i, global row m) once per lane for its 4 DMA rows;
// src(i, kstep, logical 16-B chunk) -> per-lane global address of 8 bf16.
struct DenseA {
    const bf16_t* A;
    long lda;
    int M;
    const bf16_t* rowp[4];
    __device__ __forceinline__ void init(int i, long m) { rowp[i] = A + (size_t)(m < M ? m : M - 1) * lda; }
    __device__ __forceinline__ const void* src(int i, int kstep, int chunk) const {
        return rowp[i] + kstep * BK + chunk * 8;
    }
};

// Implicit GEMM for a 3x3, stride 1, pad 1 convolution on an NHWC bf16 map with C % 64 == 0:
// row m = flat output pixel, K index = tap*C + c, tap = (dy+1)*3 + (dx+1).
struct Conv3x3A {
    const bf16_t* in;
    int H, W, C;
    long M;
    int cblocks;  // C / 64
    const bf16_t* pix[4];
    int yy[4], xx[4];
    __device__ __forceinline__ void init(int i, long m) {
        if (m >= M) m = M - 1;
        const unsigned hw = (unsigned)H * (unsigned)W;  // M < 2^31 is checked on the host
        const unsigned rem = (unsigned)m % hw;
        yy[i] = (int)(rem / (unsigned)W);
        xx[i] = (int)(rem % (unsigned)W);
        pix[i] = in + (size_t)m * C;
    }
    __device__ __forceinline__ const void* src(int i, int kstep, int chunk) const {
        const int tap = kstep / cblocks, cb = kstep - tap * cblocks;
        const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
        const bool ok = (unsigned)(yy[i] + dy) < (unsigned)H && (unsigned)(xx[i] + dx) < (unsigned)W;
        const bf16_t* p = pix[i] + ((long)dy * W + dx) * C + cb * BK + chunk * 8;
        return ok ? (const void*)p : (const void*)g_isp_zero16;
    }
};

// ------------------------------------------------------------------------------ epilogues
// Epilogue contract: operator()(m, n, v[4]) for output row m < M, columns n..n+3 < N.
enum { ACT_NONE = 0, ACT_RELU = 1, ACT_GELU = 2 };

template <int ACT>
struct EpBiasActBf16 {  // out[m][n] = bf16(act(v + bias[n]))
    bf16_t* out;
    const float* bias;  // may be null
    long ldo;
    __device__ __forceinline__ void operator()(long m, int n, const float* v) const {
        float r[4];
        float4 b = bias ? *reinterpret_cast<const float4*>(bias + n) : make_float4(0, 0, 0, 0);
        r[0] = v[0] + b.x, r[1] = v[1] + b.y, r[2] = v[2] + b.z, r[3] = v[3] + b.w;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (ACT == ACT_RELU) r[j] = fmaxf(r[j], 0.f);
            if (ACT == ACT_GELU) r[j] = gelu_erf(r[j]);
        }
        *reinterpret_cast<uint2*>(out + (size_t)m * ldo + n) = make_uint2(pack2bf(r[0], r[1]), pack2bf(r[2], r[3]));
    }
};

template <int ACT>
struct EpBiasActF32 {  // out[m][n] = act(v + bias[n]) in fp32
    float* out;
    const float* bias;
    long ldo;
    __device__ __forceinline__ void operator()(long m, int n, const float* v) const {
        float4 b = bias ? *reinterpret_cast<const float4*>(bias + n) : make_float4(0, 0, 0, 0);
        float r[4] = {v[0] + b.x, v[1] + b.y, v[2] + b.z, v[3] + b.w};
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
    __device__ __forceinline__ void operator()(long m, int n, const float* v) const {
        float4 b = bias ? *reinterpret_cast<const float4*>(bias + n) : make_float4(0, 0, 0, 0);
        float4 g = gamma ? *reinterpret_cast<const float4*>(gamma + n) : make_float4(1, 1, 1, 1);
        float4* px = reinterpret_cast<float4*>(x + (size_t)m * ldx + n);
        float4 o = *px;
        o.x += g.x * (v[0] + b.x);
        o.y += g.y * (v[1] + b.y);
        o.z += g.z * (v[2] + b.z);
        o.w += g.w * (v[3] + b.w);
        *px = o;
    }
};

struct EpAxpyResBf16 {  // out = res + alpha * (v + bias), bf16 in/out (FeatUp JBUStack final fix-up)
    bf16_t* out;
    const bf16_t* res;
    const float* bias;
    float alpha;
    long ldo;
    __device__ __forceinline__ void operator()(long m, int n, const float* v) const {
        float4 b = bias ? *reinterpret_cast<const float4*>(bias + n) : make_float4(0, 0, 0, 0);
        const uint2 u = *reinterpret_cast<const uint2*>(res + (size_t)m * ldo + n);
        const float r0 = __uint_as_float(u.x << 16) + alpha * (v[0] + b.x);
        const float r1 = __uint_as_float(u.x & 0xffff0000u) + alpha * (v[1] + b.y);
        const float r2 = __uint_as_float(u.y << 16) + alpha * (v[2] + b.z);
        const float r3 = __uint_as_float(u.y & 0xffff0000u) + alpha * (v[3] + b.w);
        *reinterpret_cast<uint2*>(out + (size_t)m * ldo + n) = make_uint2(pack2bf(r0, r1), pack2bf(r2, r3));
    }
};

struct EpTokens {  // patch-embed: token row b*(T+1)+1+t gets v + bias[n] + pos[1+t][n]
    float* x;
    const float* bias;  // b_img + b_click, pre-summed
    const float* pos;   // [T+1][N] interpolated pos-embed (row 0 = cls), may be null
    int T;
    long ldx;
    __device__ __forceinline__ void operator()(long m, int n, const float* v) const {
        const long b = m / T;
        const int t = (int)(m - b * T);
        float4 bb = *reinterpret_cast<const float4*>(bias + n);
        float4 pp = pos ? *reinterpret_cast<const float4*>(pos + (size_t)(1 + t) * ldx + n) : make_float4(0, 0, 0, 0);
        *reinterpret_cast<float4*>(x + (size_t)(b * (T + 1) + 1 + t) * ldx + n) =
            make_float4(v[0] + bb.x + pp.x, v[1] + bb.y + pp.y, v[2] + bb.z + pp.z, v[3] + bb.w + pp.w);
    }
};

// ------------------------------------------------------------------------------ the engine
__device__ __forceinline__ int swz(int row, int chunk) { return chunk ^ ((row >> 1) & 7); }

template <class AL, class EP>
__global__ __launch_bounds__(256, 2) void gemm_tile_kernel(AL al, const bf16_t* __restrict__ Wt, long M, int N, int K,
                                                           int tiles_n, int nwg, EP ep) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wg = xcd_remap(blockIdx.x, nwg);
    const int tn = wg % tiles_n;
    const long tm = wg / tiles_n;
    const long m0 = tm * BM;
    const int n0 = tn * BN;

    // --- staging assignment: wave `wid` issues DMA pieces 4*wid .. 4*wid+3 of each operand
    // tile; piece q covers tile rows 8q..8q+7 (1 KiB).  Lane -> (row = 8q + lane/8, phys
    // chunk = lane%8), source chunk = swz(row, phys).
    const int lrow = lane >> 3, pchunk = lane & 7;
    int a_chunk[4];
    const bf16_t* w_src[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = (wid * 4 + i) * 8 + lrow;
        a_chunk[i] = swz(row, pchunk);
        al.init(i, m0 + row);
        const int n = n0 + row;
        w_src[i] = Wt + (size_t)(n < N ? n : N - 1) * K + a_chunk[i] * 8;
    }
    auto stage = [&](int kstep, char* buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) glds16(al.src(i, kstep, a_chunk[i]), buf + (wid * 4 + i) * 1024);
#pragma unroll
        for (int i = 0; i < 4; ++i) glds16(w_src[i] + (size_t)kstep * BK, buf + TILE_BYTES + (wid * 4 + i) * 1024);
    };

    // --- fragment read addresses (bytes inside an operand tile)
    const int wm = wid >> 1, wn = wid & 1;
    const int fr = lane & 15, fq = lane >> 4;
    int a_off[4][2], w_off[4][2];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int ra = wm * 64 + t * 16 + fr, rw = wn * 64 + t * 16 + fr;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            a_off[t][ks] = ra * 128 + swz(ra, ks * 4 + fq) * 16;
            w_off[t][ks] = TILE_BYTES + rw * 128 + swz(rw, ks * 4 + fq) * 16;
        }
    }

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto compute = [&](const char* buf) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 fa[4], fw[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                fa[t] = *reinterpret_cast<const bf16x8*>(buf + a_off[t][ks]);
                fw[t] = *reinterpret_cast<const bf16x8*>(buf + w_off[t][ks]);
            }
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[ni], fa[mi], acc[mi][ni], 0, 0, 0);
        }
    };

    const int nk = K / BK;
    char* buf0 = smem;
    char* buf1 = smem + STAGE_BYTES;
    stage(0, buf0);
    __syncthreads();
    int kt = 0;
    for (; kt + 2 <= nk; kt += 2) {  // unrolled by 2 so both LDS buffers are compile-time constants
        stage(kt + 1, buf1);
        compute(buf0);
        __syncthreads();
        if (kt + 2 < nk) stage(kt + 2, buf0);
        compute(buf1);
        __syncthreads();
    }
    if (kt < nk) compute(buf0);  // odd tail (its tile was staged by the last loop iteration / prologue)

    // --- epilogue: lane holds, for mi/ni, row m = .. + fr and 4 consecutive n = .. + 4*fq + j
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
        const long m = m0 + wm * 64 + mi * 16 + fr;
        if (m >= M) continue;
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            const int n = n0 + wn * 64 + ni * 16 + fq * 4;
            if (n >= N) continue;
            const float v[4] = {acc[mi][ni][0], acc[mi][ni][1], acc[mi][ni][2], acc[mi][ni][3]};
            ep(m, n, v);
        }
    }
}

template <class AL, class EP>
int launch_gemm(AL al, const void* Wt, long M, int N, int K, EP ep, hipStream_t s) {
    if (M <= 0 || N <= 0 || K <= 0 || K % BK != 0 || N % 4 != 0) return ISP_ERR_INVALID;
    const long tiles_m = (M + BM - 1) / BM;
    const int tiles_n = (N + BN - 1) / BN;
    const long nwg = tiles_m * tiles_n;
    if (nwg > 0x7fffffffL) return ISP_ERR_INVALID;
    static bool attr_done = false;  // per instantiation; raising the dynamic-LDS cap is idempotent
    auto kern = gemm_tile_kernel<AL, EP>;
    if (!attr_done) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) != hipSuccess)
            return ISP_ERR_LAUNCH;
        attr_done = true;
    }
    kern<<<(unsigned)nwg, 256, LDS_BYTES, s>>>(al, (const bf16_t*)Wt, M, N, K, tiles_n, (int)nwg, ep);
    return isp_launch_status();
}

template <class AL>
int dispatch_epilogue(AL al, const void* Wt, long M, int N, int K, const isp_epilogue* e, hipStream_t s) {
    if (!e || !e->out) return ISP_ERR_INVALID;
    const long ldo = e->ldo > 0 ? e->ldo : N;
    if (ldo % 4 != 0) return ISP_ERR_INVALID;
    switch (e->kind) {
        case ISP_EP_BIAS_BF16:
            return launch_gemm(al, Wt, M, N, K, EpBiasActBf16<ACT_NONE>{(bf16_t*)e->out, e->bias, ldo}, s);
        case ISP_EP_BIAS_RELU_BF16:
            return launch_gemm(al, Wt, M, N, K, EpBiasActBf16<ACT_RELU>{(bf16_t*)e->out, e->bias, ldo}, s);
        case ISP_EP_BIAS_GELU_BF16:
            return launch_gemm(al, Wt, M, N, K, EpBiasActBf16<ACT_GELU>{(bf16_t*)e->out, e->bias, ldo}, s);
        case ISP_EP_BIAS_F32:
            return launch_gemm(al, Wt, M, N, K, EpBiasActF32<ACT_NONE>{(float*)e->out, e->bias, ldo}, s);
        case ISP_EP_RESIDUAL_F32:
            return launch_gemm(al, Wt, M, N, K, EpResidual{(float*)e->out, e->bias, e->gamma, ldo}, s);
        case ISP_EP_TOKENS_F32:
            if (e->tokens_per_image <= 0 || !e->bias) return ISP_ERR_INVALID;
            return launch_gemm(al, Wt, M, N, K, EpTokens{(float*)e->out, e->bias, e->pos, e->tokens_per_image, ldo}, s);
        case ISP_EP_AXPY_RES_BF16:
            if (!e->res) return ISP_ERR_INVALID;
            return launch_gemm(al, Wt, M, N, K,
                               EpAxpyResBf16{(bf16_t*)e->out, (const bf16_t*)e->res, e->bias, e->alpha, ldo}, s);
        default:
            return ISP_ERR_UNSUPPORTED;
    }
}

}  // namespace

extern "C" int isp_gemm_bf16(const void* A, long lda, const void* Wt, long M, int N, int K, const isp_epilogue* ep,
                             void* stream) {
    ISP_CHECK_ARG(A && Wt && lda >= K && lda % 8 == 0);
    DenseA al;
    al.A = (const bf16_t*)A;
    al.lda = lda;
    al.M = (int)M;
    ISP_CHECK_ARG(M <= 0x7fffffffL);
    return dispatch_epilogue(al, Wt, M, N, K, ep, (hipStream_t)stream);
}

extern "C" int isp_conv3x3_nhwc_bf16(const void* in, const void* Wt, int B, int H, int W, int C, int N,
                                     const isp_epilogue* ep, void* stream) {
    ISP_CHECK_ARG(in && Wt && B > 0 && H > 0 && W > 0 && C > 0 && C % BK == 0);
    Conv3x3A al;
    al.in = (const bf16_t*)in;
    al.H = H;
    al.W = W;
    al.C = C;
    al.M = (long)B * H * W;
    ISP_CHECK_ARG(al.M <= 0x7fffffffL);
    al.cblocks = C / BK;
    return dispatch_epilogue(al, Wt, al.M, N, 9 * C, ep, (hipStream_t)stream);
}
