// Fused softmax(Q K^T * scale) V on gfx950 for head_dim 64 and 128 (bf16 MFMA 32x32x16, fp32
// online softmax); the [Lq, Lk] score matrix never leaves registers.
//
// Replaces Attention.forward (reference dinov2/layers/attention.py:54-71, head_dim 64), which
// materialises the [B, heads, N, N] scores in HBM, and nn.MultiheadAttention inside LoftUp's
// CrossAttentionLayer (loftup/layers.py:182-198; head_dim 101 zero-padded to 128), which
// materialises a 3.3 GB attention-weight tensor per image and layer.
//
// Structure (per block: 4 waves x 32 queries of one (batch, head); KV tiles of 64 keys):
//   * K and V tiles arrive by 16-byte LDS-DMA into a 2-deep LDS ring;
//   * S^T = K . Q^T ("swapped" product): a lane owns ONE query (column) and 16 of the 32
//     keys of each key block, so the row max / row sum are in-lane + one lane^32 exchange;
//   * the S^T accumulator, converted to bf16 in place, IS the B operand of the PV product
//     O^T += V^T . P^T (accumulator-as-operand idiom): no LDS round trip for P;
//   * V^T fragments come from the row-major V tile through ds_read_b64_tr_b16;
//   * 16-byte chunks of the K rows are XOR-swizzled so the ds_read_b128 fragment reads are
//     conflict-free, those of the V rows so the transposed reads are; both swizzles are applied
//     on the DMA source address, the LDS destination stays lane-linear.
#include "isp_common.h"

namespace {

// queries per block = 32 per wave; NW = 4 waves (128 queries), or 8 (256) for very long query sequences with short
// key sequences (LoftUp: 200 k pixels x 1 k keys), where every block streams ALL keys of its (batch, head) through LDS
// and the kernel sits on the L2 -> LDS staging rate: twice the queries per streamed key byte.
constexpr int KB = 64;   // keys per tile

typedef __attribute__((ext_vector_type(4))) short s16x4;

__device__ __forceinline__ s16x4 tr_read(const char* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((ISP_LDS s16x4*)p);
}

template <int HD>
struct Geo {
    static constexpr int ROW = HD * 2;             // bytes per key row
    static constexpr int CHUNKS = ROW / 16;        // 16-byte chunks per row (8 or 16)
    static constexpr int TILE = KB * ROW;          // bytes per K (or V) tile
    static constexpr int ROWS_PER_PIECE = 1024 / ROW;  // rows covered by one 1 KiB DMA piece
    static constexpr int PIECES = TILE / 1024;     // per operand tile
    static constexpr int LDS = 4 * TILE;           // (K + V) x 2 buffers
    // 128-byte rows: 16 consecutive rows share 2 bank-row positions -> spread with (row>>1)&7;
    // 256-byte rows: every row starts on the same bank -> spread with row&15.
    __device__ static __forceinline__ int kswz(int row, int chunk) {
        return HD == 64 ? chunk ^ ((row >> 1) & 7) : chunk ^ (row & 15);  // 256- and 512-byte rows alike
    }
    // transposed reads touch 4 consecutive key rows x 64 contiguous bytes per half-wave
    __device__ static __forceinline__ int vswz(int row, int chunk) {
        return HD == 64 ? chunk ^ (((row >> 1) & 1) << 2) : chunk ^ ((row & 3) << 2);
    }
};

template <int HD, int NW>
__global__ __launch_bounds__(64 * NW) void attention_kernel(const bf16_t* __restrict__ Q, const bf16_t* __restrict__ K,
                                                        const bf16_t* __restrict__ V, bf16_t* __restrict__ O, int H,
                                                        int Lq, int Lk, long qsb, long qsl, long qsh, long ksb, long ksl,
                                                        long ksh, long osb, long osl, long osh, float c /* scale*log2e */,
                                                        float* __restrict__ lse, long lse_ld) {
    using G = Geo<HD>;
    constexpr int KK = HD / 16;  // k-steps of the QK^T product
    constexpr int DB = HD / 32;  // 32-wide blocks of the head dim (rows of O^T)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.y / H, h = blockIdx.y % H;
    const int r = lane & 31, hh = lane >> 5;

    // ---- Q fragments (B operand of S^T = K Q^T): element j <-> d = 16kk + 8hh + j
    constexpr int QB = 32 * NW;
    const long qrow = (long)blockIdx.x * QB + wid * 32 + r;
    const bf16_t* qp = Q + (size_t)b * qsb + (size_t)(qrow < Lq ? qrow : Lq - 1) * qsl + (size_t)h * qsh + 8 * hh;
    bf16x8 qf[KK];
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) qf[kk] = *reinterpret_cast<const bf16x8*>(qp + 16 * kk);

    // ---- DMA assignment: each operand tile = PIECES pieces of 1 KiB; wave w takes pieces w, w+4, ...
    const bf16_t* kbase = K + (size_t)b * ksb + (size_t)h * ksh;
    const bf16_t* vbase = V + (size_t)b * ksb + (size_t)h * ksh;
    static_assert(G::PIECES % NW == 0);
    constexpr int PPW = G::PIECES / NW;
    int prow[PPW], kch[PPW], vch[PPW];
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        const int piece = wid + NW * i;
        prow[i] = piece * G::ROWS_PER_PIECE + lane / G::CHUNKS;
        const int pch = lane % G::CHUNKS;
        kch[i] = G::kswz(prow[i], pch);
        vch[i] = G::vswz(prow[i], pch);
    }
    auto stage = [&](int tile, char* buf) {
        const int base = tile * KB;
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            int key = base + prow[i];
            key = key < Lk ? key : Lk - 1;
            glds16(kbase + (size_t)key * ksl + kch[i] * 8, buf + (wid + NW * i) * 1024);
            glds16(vbase + (size_t)key * ksl + vch[i] * 8, buf + G::TILE + (wid + NW * i) * 1024);
        }
    };

    // ---- fragment addresses
    int k_off[2][KK];  // [kb][kk]: K row kb*32 + r, logical chunk 2kk + hh
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
        const int row = kb * 32 + r;
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) k_off[kb][kk] = row * G::ROW + (G::kswz(row, 2 * kk + hh) << 4);
    }
    // V^T fragment via transposed reads: 16-lane group g: d0 = db*32 + 16*(g&1), keys +4*hh;
    // lane i=4q+p of the group supplies &V[key0+q][d0+4p].  Offsets for db = 0; db adds 64 bytes
    // of logical column, i.e. 4 chunks: folded in through vswz below.
    const int gi = lane & 15, gq = gi >> 2, gp = gi & 3, g1 = (lane >> 4) & 1;
    int v_off[DB][2][2][2];  // [db][kb][s][jj]
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    const int key = kb * 32 + 16 * s + 8 * jj + 4 * hh + gq;
                    const int col = db * 32 + 16 * g1 + 4 * gp;
                    v_off[db][kb][s][jj] = G::TILE + key * G::ROW + G::vswz(key, col >> 3) * 16 + (col & 7) * 2;
                }

    f32x16 o[DB];
#pragma unroll
    for (int d = 0; d < DB; ++d)
#pragma unroll
        for (int i = 0; i < 16; ++i) o[d][i] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    const int nt = (Lk + KB - 1) / KB;
    stage(0, smem);
    __syncthreads();
    for (int t = 0; t < nt; ++t) {
        const char* buf = (t & 1) ? smem + 2 * G::TILE : smem;
        if (t + 1 < nt) stage(t + 1, (t & 1) ? smem : smem + 2 * G::TILE);

        // S^T = K Q^T
        f32x16 s[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int i = 0; i < 16; ++i) s[kb][i] = 0.f;
#pragma unroll
            for (int kk = 0; kk < KK; ++kk) {
                const bf16x8 kf = *reinterpret_cast<const bf16x8*>(buf + k_off[kb][kk]);
                s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[kk], s[kb], 0, 0, 0);
            }
        }
        // mask keys past Lk (last tile only; wave-uniform branch)
        const int kbase_idx = t * KB;
        if (kbase_idx + KB > Lk) {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int key = kbase_idx + kb * 32 + (i & 3) + 8 * (i >> 2) + 4 * hh;
                    if (key >= Lk) s[kb][i] = -INFINITY;
                }
        }
        // online softmax (base-2)
        float mx = s[0][0];
#pragma unroll
        for (int i = 1; i < 16; ++i) mx = fmaxf(mx, s[0][i]);
#pragma unroll
        for (int i = 0; i < 16; ++i) mx = fmaxf(mx, s[1][i]);
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float m_new = fmaxf(m_run, mx * c);
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        m_run = m_new;
        float psum = 0.f;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float p = __builtin_amdgcn_exp2f(fmaf(s[kb][i], c, -m_new));  // (explicit fma: -ffp-contract=off)
                s[kb][i] = p;
                psum += p;
            }
        l_run = l_run * alpha + psum;
#pragma unroll
        for (int d = 0; d < DB; ++d)
#pragma unroll
            for (int i = 0; i < 16; ++i) o[d][i] *= alpha;

        // O^T += V^T P^T
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int ss = 0; ss < 2; ++ss) {
                bf16x8 pf;
#pragma unroll
                for (int j = 0; j < 8; ++j) pf[j] = (short)f2bf(s[kb][8 * ss + j]);
#pragma unroll
                for (int db = 0; db < DB; ++db) {
                    const s16x4 lo = tr_read(buf + v_off[db][kb][ss][0]);
                    const s16x4 hi = tr_read(buf + v_off[db][kb][ss][1]);
                    const bf16x8 vf = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, o[db], 0, 0, 0);
                }
            }
        __syncthreads();
    }

    // ---- epilogue: O[b, q, h, d] = o / l
    const float l_tot = l_run + __shfl_xor(l_run, 32);
    const float inv = 1.0f / l_tot;
    // base-2 log-sum-exp of the scaled scores, kept for the backward (P = exp2(s*c - lse))
    if (lse && hh == 0 && qrow < Lq) lse[(size_t)blockIdx.y * lse_ld + qrow] = m_run + log2f(l_tot);
    if (qrow < Lq) {
        bf16_t* op = O + (size_t)b * osb + (size_t)qrow * osl + (size_t)h * osh;
#pragma unroll
        for (int db = 0; db < DB; ++db)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int d = db * 32 + 8 * g4 + 4 * hh;
                *reinterpret_cast<uint2*>(op + d) =
                    make_uint2(pack2bf(o[db][4 * g4 + 0] * inv, o[db][4 * g4 + 1] * inv),
                               pack2bf(o[db][4 * g4 + 2] * inv, o[db][4 * g4 + 3] * inv));
            }
    }
}

template <int HD, int NW = 4>
int launch_attention(const void* Q, const void* K, const void* V, void* O, int B, int H, int Lq, int Lk, long qsb,
                     long qsl, long qsh, long ksb, long ksl, long ksh, long osb, long osl, long osh, float scale,
                     float* lse, long lse_ld, hipStream_t s) {
    static bool attr_done = false;
    constexpr int QB = 32 * NW;
    auto kern = attention_kernel<HD, NW>;
    if (!attr_done) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, Geo<HD>::LDS) != hipSuccess)
            return ISP_ERR_LAUNCH;
        attr_done = true;
    }
    dim3 grid((Lq + QB - 1) / QB, B * H);
    kern<<<grid, 64 * NW, Geo<HD>::LDS, s>>>((const bf16_t*)Q, (const bf16_t*)K, (const bf16_t*)V, (bf16_t*)O, H, Lq, Lk, qsb,
                                         qsl, qsh, ksb, ksl, ksh, osb, osl, osh, scale * 1.4426950408889634f, lse,
                                         lse_ld);
    return isp_launch_status();
}

}  // namespace

static int attention_fwd_impl(const void* Q, const void* K, const void* V, void* O, int B, int H, int Lq, int Lk,
                              int head_dim, long q_stride_b, long q_stride_l, long q_stride_h, long kv_stride_b,
                              long kv_stride_l, long kv_stride_h, long o_stride_b, long o_stride_l, long o_stride_h,
                              float scale, float* lse, long lse_ld, void* stream) {
    ISP_CHECK_ARG(Q && K && V && O && B > 0 && H > 0 && Lq > 0 && Lk > 0 && scale > 0.f);
    ISP_CHECK_ARG((long)B * H <= 65535);
    // 16-byte vector loads / 8-byte stores need aligned strides
    ISP_CHECK_ARG(q_stride_b % 8 == 0 && q_stride_l % 8 == 0 && q_stride_h % 8 == 0);
    ISP_CHECK_ARG(kv_stride_b % 8 == 0 && kv_stride_l % 8 == 0 && kv_stride_h % 8 == 0);
    ISP_CHECK_ARG(o_stride_b % 4 == 0 && o_stride_l % 4 == 0 && o_stride_h % 4 == 0);
    ISP_CHECK_ARG(!lse || lse_ld >= Lq);
    hipStream_t s = (hipStream_t)stream;
    if (head_dim == 64)
        return launch_attention<64>(Q, K, V, O, B, H, Lq, Lk, q_stride_b, q_stride_l, q_stride_h, kv_stride_b,
                                    kv_stride_l, kv_stride_h, o_stride_b, o_stride_l, o_stride_h, scale, lse, lse_ld, s);
    if (head_dim == 128) {
        // 256-query blocks once they still fill the chip several times over
        if ((long)((Lq + 255) / 256) * B * H >= 2048)
            return launch_attention<128, 8>(Q, K, V, O, B, H, Lq, Lk, q_stride_b, q_stride_l, q_stride_h, kv_stride_b,
                                            kv_stride_l, kv_stride_h, o_stride_b, o_stride_l, o_stride_h, scale, lse, lse_ld, s);
        return launch_attention<128>(Q, K, V, O, B, H, Lq, Lk, q_stride_b, q_stride_l, q_stride_h, kv_stride_b,
                                     kv_stride_l, kv_stride_h, o_stride_b, o_stride_l, o_stride_h, scale, lse, lse_ld, s);
    }
    if (head_dim == 256) {
        if ((long)((Lq + 255) / 256) * B * H >= 2048)
            return launch_attention<256, 8>(Q, K, V, O, B, H, Lq, Lk, q_stride_b, q_stride_l, q_stride_h, kv_stride_b,
                                            kv_stride_l, kv_stride_h, o_stride_b, o_stride_l, o_stride_h, scale, lse, lse_ld, s);
        return launch_attention<256>(Q, K, V, O, B, H, Lq, Lk, q_stride_b, q_stride_l, q_stride_h, kv_stride_b,
                                     kv_stride_l, kv_stride_h, o_stride_b, o_stride_l, o_stride_h, scale, lse, lse_ld, s);
    }
    return ISP_ERR_UNSUPPORTED;
}

extern "C" int isp_attention_fwd(const void* Q, const void* K, const void* V, void* O, int B, int H, int Lq, int Lk,
                                 int head_dim, long q_stride_b, long q_stride_l, long q_stride_h, long kv_stride_b,
                                 long kv_stride_l, long kv_stride_h, long o_stride_b, long o_stride_l,
                                 long o_stride_h, float scale, void* stream) {
    return attention_fwd_impl(Q, K, V, O, B, H, Lq, Lk, head_dim, q_stride_b, q_stride_l, q_stride_h, kv_stride_b,
                              kv_stride_l, kv_stride_h, o_stride_b, o_stride_l, o_stride_h, scale, nullptr, 0, stream);
}

extern "C" int isp_attention_fwd_lse(const void* Q, const void* K, const void* V, void* O, float* lse, long lse_ld, int B,
                                     int H, int Lq, int Lk, int head_dim, long q_stride_b, long q_stride_l,
                                     long q_stride_h, long kv_stride_b, long kv_stride_l, long kv_stride_h,
                                     long o_stride_b, long o_stride_l, long o_stride_h, float scale, void* stream) {
    ISP_CHECK_ARG(lse);
    return attention_fwd_impl(Q, K, V, O, B, H, Lq, Lk, head_dim, q_stride_b, q_stride_l, q_stride_h, kv_stride_b,
                              kv_stride_l, kv_stride_h, o_stride_b, o_stride_l, o_stride_h, scale, lse, lse_ld, stream);
}
