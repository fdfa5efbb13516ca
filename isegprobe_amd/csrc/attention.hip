// Fused softmax(Q K^T * scale) V for head_dim 64 on gfx950 (bf16 MFMA 32x32x16, fp32
// online softmax); the [Lq, Lk] score matrix never leaves registers.
//
// Replaces Attention.forward (reference dinov2/layers/attention.py:54-71), which
// materialises the [B, heads, N, N] scores in HBM.
//
// Structure (per block: 4 waves x 32 queries of one (batch, head); KV tiles of 64 keys):
//   * K and V tiles arrive by 16-byte LDS-DMA into a 2-deep LDS ring (32 KiB);
//   * S^T = K . Q^T ("swapped" product): a lane owns ONE query (column) and 16 of the 32
//     keys of each key block, so the row max / row sum are in-lane + one lane^32 exchange;
//   * the S^T accumulator, converted to bf16 in place, IS the B operand of the PV product
//     O^T += V^T . P^T (accumulator-as-operand idiom): no LDS round trip for P;
//   * V^T fragments come from the row-major V tile through ds_read_b64_tr_b16;
//   * K tile chunks are XOR-swizzled with (row>>1)&7 (conflict-free ds_read_b128), V tile
//     chunks with ((row>>1)&1)<<2 (conflict-free transposed reads); both swizzles are
//     applied on the DMA source address, the LDS destination stays lane-linear.
#include "isp_common.h"

namespace {

constexpr int QB = 128;  // queries per block
constexpr int KB = 64;   // keys per tile
constexpr int HD = 64;   // head dim
constexpr int KV_TILE = KB * HD * 2;  // 8 KiB
constexpr int ATT_LDS = 4 * KV_TILE;  // (K + V) x 2 buffers

typedef __attribute__((ext_vector_type(4))) short s16x4;

__device__ __forceinline__ s16x4 tr_read(const char* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((ISP_LDS s16x4*)p);
}

__global__ __launch_bounds__(256) void attention_kernel(const bf16_t* __restrict__ Q, const bf16_t* __restrict__ K,
                                                        const bf16_t* __restrict__ V, bf16_t* __restrict__ O, int H,
                                                        int Lq, int Lk, long qsb, long qsl, long qsh, long ksb, long ksl,
                                                        long ksh, long osb, long osl, long osh, float c /* scale*log2e */) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.y / H, h = blockIdx.y % H;
    const int r = lane & 31, hh = lane >> 5;

    // ---- Q fragments (B operand of S^T = K Q^T): element j <-> d = 16kk + 8hh + j
    const int qrow = blockIdx.x * QB + wid * 32 + r;
    const bf16_t* qp = Q + (size_t)b * qsb + (size_t)(qrow < Lq ? qrow : Lq - 1) * qsl + (size_t)h * qsh + 8 * hh;
    bf16x8 qf[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) qf[kk] = *reinterpret_cast<const bf16x8*>(qp + 16 * kk);

    // ---- DMA assignment: K tile = 8 pieces of 8 rows; wave w takes K pieces 2w,2w+1 and V pieces 2w,2w+1
    const bf16_t* kbase = K + (size_t)b * ksb + (size_t)h * ksh;
    const bf16_t* vbase = V + (size_t)b * ksb + (size_t)h * ksh;
    const int lrow = lane >> 3, pch = lane & 7;
    int krow[2], kch[2], vch[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        krow[i] = (wid * 2 + i) * 8 + lrow;
        kch[i] = pch ^ ((krow[i] >> 1) & 7);
        vch[i] = pch ^ (((krow[i] >> 1) & 1) << 2);
    }
    auto stage = [&](int tile, char* buf) {
        const int base = tile * KB;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            int key = base + krow[i];
            key = key < Lk ? key : Lk - 1;
            glds16(kbase + (size_t)key * ksl + kch[i] * 8, buf + (wid * 2 + i) * 1024);
            glds16(vbase + (size_t)key * ksl + vch[i] * 8, buf + KV_TILE + (wid * 2 + i) * 1024);
        }
    };

    // ---- fragment addresses
    int k_off[2][4];  // [kb][kk]: K row kb*32 + r, logical chunk 2kk + hh
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
        const int row = kb * 32 + r;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) k_off[kb][kk] = row * 128 + (((2 * kk + hh) ^ ((row >> 1) & 7)) << 4);
    }
    // V^T fragment via transposed reads: 16-lane group g: d0 = db*32 + 16*(g&1), keys +4*hh;
    // lane i=4q+p of the group supplies &V[key0+q][d0+4p]
    const int gi = lane & 15, gq = gi >> 2, gp = gi & 3, g1 = (lane >> 4) & 1;
    int v_off[2][2][2][2];  // [db][kb][s][jj]
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    const int key = kb * 32 + 16 * s + 8 * jj + 4 * hh + gq;
                    const int col = db * 32 + 16 * g1 + 4 * gp;
                    const int chunk = (col >> 3) ^ (((key >> 1) & 1) << 2);
                    v_off[db][kb][s][jj] = KV_TILE + key * 128 + chunk * 16 + (col & 7) * 2;
                }

    f32x16 o[2];
#pragma unroll
    for (int i = 0; i < 16; ++i) o[0][i] = 0.f, o[1][i] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    const int nt = (Lk + KB - 1) / KB;
    stage(0, smem);
    __syncthreads();
    for (int t = 0; t < nt; ++t) {
        const char* buf = (t & 1) ? smem + 2 * KV_TILE : smem;
        if (t + 1 < nt) stage(t + 1, (t & 1) ? smem : smem + 2 * KV_TILE);

        // S^T = K Q^T
        f32x16 s[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int i = 0; i < 16; ++i) s[kb][i] = 0.f;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
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
                const float p = __builtin_amdgcn_exp2f(s[kb][i] * c - m_new);
                s[kb][i] = p;
                psum += p;
            }
        l_run = l_run * alpha + psum;
#pragma unroll
        for (int i = 0; i < 16; ++i) o[0][i] *= alpha, o[1][i] *= alpha;

        // O^T += V^T P^T
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int ss = 0; ss < 2; ++ss) {
                bf16x8 pf;
#pragma unroll
                for (int j = 0; j < 8; ++j) pf[j] = (short)f2bf(s[kb][8 * ss + j]);
#pragma unroll
                for (int db = 0; db < 2; ++db) {
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
    if (qrow < Lq) {
        bf16_t* op = O + (size_t)b * osb + (size_t)qrow * osl + (size_t)h * osh;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int d = db * 32 + 8 * g4 + 4 * hh;
                *reinterpret_cast<uint2*>(op + d) =
                    make_uint2(pack2bf(o[db][4 * g4 + 0] * inv, o[db][4 * g4 + 1] * inv),
                               pack2bf(o[db][4 * g4 + 2] * inv, o[db][4 * g4 + 3] * inv));
            }
    }
}

}  // namespace

extern "C" int isp_attention_fwd(const void* Q, const void* K, const void* V, void* O, int B, int H, int Lq, int Lk,
                                 long q_stride_b, long q_stride_l, long q_stride_h, long kv_stride_b, long kv_stride_l,
                                 long kv_stride_h, long o_stride_b, long o_stride_l, long o_stride_h, float scale,
                                 void* stream) {
    ISP_CHECK_ARG(Q && K && V && O && B > 0 && H > 0 && Lq > 0 && Lk > 0 && scale > 0.f);
    ISP_CHECK_ARG((long)B * H <= 65535);
    // 16-byte vector loads / 8-byte stores need aligned strides
    ISP_CHECK_ARG(q_stride_b % 8 == 0 && q_stride_l % 8 == 0 && q_stride_h % 8 == 0);
    ISP_CHECK_ARG(kv_stride_b % 8 == 0 && kv_stride_l % 8 == 0 && kv_stride_h % 8 == 0);
    ISP_CHECK_ARG(o_stride_b % 4 == 0 && o_stride_l % 4 == 0 && o_stride_h % 4 == 0);
    dim3 grid((Lq + QB - 1) / QB, B * H);
    attention_kernel<<<grid, 256, ATT_LDS, (hipStream_t)stream>>>(
        (const bf16_t*)Q, (const bf16_t*)K, (const bf16_t*)V, (bf16_t*)O, H, Lq, Lk, q_stride_b, q_stride_l,
        q_stride_h, kv_stride_b, kv_stride_l, kv_stride_h, o_stride_b, o_stride_l, o_stride_h,
        scale * 1.4426950408889634f);
    return isp_launch_status();
}
