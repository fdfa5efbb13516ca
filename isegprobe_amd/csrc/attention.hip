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

template <int HD, int NW, bool F16 = false>  // F16: Q, K, V, P and O are IEEE half (isp_attention_fwd_f16)
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
                if constexpr (F16)
                    s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, kf), __builtin_bit_cast(f16x8_t, qf[kk]),
                                                                  s[kb], 0, 0, 0);
                else
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
                for (int j = 0; j < 8; j += 2) {
                    const unsigned pk = pack2o<!F16>(s[kb][8 * ss + j], s[kb][8 * ss + j + 1]);
                    pf[j] = (short)(pk & 0xffffu), pf[j + 1] = (short)(pk >> 16);
                }
#pragma unroll
                for (int db = 0; db < DB; ++db) {
                    const s16x4 lo = tr_read(buf + v_off[db][kb][ss][0]);
                    const s16x4 hi = tr_read(buf + v_off[db][kb][ss][1]);
                    const bf16x8 vf = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    if constexpr (F16)
                        o[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, vf), __builtin_bit_cast(f16x8_t, pf),
                                                                      o[db], 0, 0, 0);
                    else
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
                    make_uint2(pack2o_sat<!F16>(o[db][4 * g4 + 0] * inv, o[db][4 * g4 + 1] * inv),
                               pack2o_sat<!F16>(o[db][4 * g4 + 2] * inv, o[db][4 * g4 + 3] * inv));
            }
    }
}

template <int HD, int NW = 4, bool F16 = false>
int launch_attention(const void* Q, const void* K, const void* V, void* O, int B, int H, int Lq, int Lk, long qsb,
                     long qsl, long qsh, long ksb, long ksl, long ksh, long osb, long osl, long osh, float scale,
                     float* lse, long lse_ld, hipStream_t s) {
    static bool attr_done = false;
    constexpr int QB = 32 * NW;
    auto kern = attention_kernel<HD, NW, F16>;
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


// ---------------------------------------------------------------------------------------------------------------
// head_dim 64 (the ViT's self-attention): the generic kernel above spends 3x its MFMA time in VALU work at this head
// size (per 64-key tile and wave: 16 MFMAs = 512 cycles, against ~280 vector instructions + 32 v_exp), so this variant
// removes vector instructions instead of re-tiling:
//   * the MFMA chain's initial accumulator is -m (the row's reference maximum, a 16-register constant per lane): no
//     accumulator zeroing and no per-element subtraction; with PRESCALED (isp_attention_fwd_logit2: Q already carries
//     scale*log2(e), the ViT folds it into the Q rows of its qkv weights) the chain's output goes straight into v_exp,
//     otherwise one v_mul per element remains (re-rounding Q*c to bf16 inside the kernel would avoid it but changes the
//     scores by up to 2^-9 of their largest term: visible when |score| >> 1);
//   * m is only moved when some row's tile maximum exceeds it by more than ATT_THR (base-2 units): the O / l rescale,
//     the refresh of the -m registers and the correction of the tile in flight then sit behind a wave-uniform branch
//     that is taken for the first tile and rarely afterwards (P <= 2^ATT_THR in between; the decision precedes the
//     tile's exponentials, so nothing at the old reference is left unscaled);
//   * the row sums come out of the PV product: a third A operand of ones accumulates l = sum_k bf16(P) in 16 more
//     accumulator registers (4 MFMAs per tile instead of 32 adds; the normaliser then matches the rounded numerator);
//   * 256 registers per wave at most (two blocks per CU), so the accumulators live in VGPRs and the rescale needs no
//     AGPR copies; K/V tiles by buffer_load ... lds against SGPR resources whose range ends at the last key (rows
//     past it read zeros): no per-tile address arithmetic; the cross-half maximum by v_permlane32_swap.
// Waves whose 32 queries are all past Lq only stage tiles; when the last tile holds <= 32 keys its second key block
// is skipped (L = 1025: one block row and one tile per (batch, head) exist for the class token alone).
// Measured at B 32 x 6 heads x L 1025 (51.6 GFLOP): 86 us against 101-111 us for the generic kernel.  The PMC passes
// (tools/att_pmc.sh) put the VALU at 44 % and the MFMA pipe at 37 % of the kernel's cycles with no LDS bank conflicts:
// what is left is latency, and three waves per SIMD hide it best -- a software-pipelined variant at two waves per SIMD
// (K fragments and V reads a phase ahead, three LDS slots; 96 us), the same with register-staged tiles (108 us) and
// this kernel squeezed to four waves per SIMD (101 us) all lost.  Removing the exponentials, the PV product, the V
// reads or the barrier from the loop moves the time by < 12 % each.  Two more ablations locate the rest: without the
// whole softmax (max, rebase, exp) the time does not move (90.8 vs 91.1 us: the VALU work is fully hidden), and with every
// MFMA replaced by one VALU op it drops to 51.6 us -- a 52 us skeleton of LDS fragment reads (each of the block's four
// waves reads ALL of a tile's K and V fragments: 1.9 GB of LDS reads per launch, ~27 us at 128 B/clk/CU, plus 7 us of DMA
// writes), barriers and stores, with ~40 us of MFMA time (30 us at a full pipe) ADDED to it rather than hidden under it.
// What would move it: 64 queries (two 32-query streams) per wave sharing every K / V fragment read -- half the LDS
// traffic per query -- at ~220 registers (no -m accumulator trick, VALU row sums: the VALU has the room).
#ifndef ISP_ATT_THR
#define ISP_ATT_THR 6.0f
#endif
#ifndef ISP_ATT_ONES
#define ISP_ATT_ONES (-1)  // -1: by head_dim (see attention64_body)
#endif

__device__ __forceinline__ float xhalf_max(float x) {  // max(x of lane, x of lane ^ 32)
    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
}

template <bool F16>
__device__ __forceinline__ f32x16 att_mfma(const bf16x8& a, const bf16x8& b, const f32x16& c) {
    if constexpr (F16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// HD / NW: written for head_dim 64 with 4 waves (the ViT); instantiated as well for head_dim 128 with 8 waves = 256 queries
// per workgroup (LoftUp's cross-attention, 200 k pixel queries x 1 k keys: half the K / V bytes staged per query), where
// the deferred maximum removes the per-tile rescale of 64 accumulator registers the generic kernel pays.
template <bool PRESCALED, bool F16 = false, int HD = 64, int NW_ = 4>  // F16: Q, K, V, P and O are IEEE half
__device__ __forceinline__ void attention64_body(const bf16_t* __restrict__ Q, const bf16_t* __restrict__ K,
                                                 const bf16_t* __restrict__ V, bf16_t* __restrict__ O, int H, int Lq, int Lk,
                                                 long qsb, long qsl, long qsh, long ksb, long ksl, long ksh, long osb, long osl,
                                                 long osh, float c_arg /* scale*log2e */, float* __restrict__ lse, long lse_ld,
                                                 int nbh, int nqb) {
    using G = Geo<HD>;
    const float c = PRESCALED ? 1.f : c_arg;  // scores, m_run and the threshold are in units of 1/c base-2 logits
    const float thr = PRESCALED ? ISP_ATT_THR : ISP_ATT_THR / c_arg;
    constexpr int NW = NW_, KK = HD / 16, DB = HD / 32, QB = 32 * NW, PPW = G::PIECES / NW;  // 2 K + 2 V pieces per wave and tile
    static_assert(G::PIECES % NW == 0, "pieces per wave");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    // Workgroups go to the 8 XCDs round-robin by linear id, and each XCD has its own L2: with the q-blocks of one
    // (batch, head) on consecutive ids, its K and V (262 KB at L = 1025) are pulled over the fabric into all 8 L2s --
    // 400 MB per launch at B = 32, which alone takes the ~90 us the kernel ran in.  Remapped so that the q-blocks of a
    // (batch, head) share id % 8 (same XCD, dispatched back to back) whenever the number of (batch, head) pairs allows.
    int bh, qblk;
    if (nbh % 8 == 0) {
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        bh = (slot / nqb) * 8 + xcd, qblk = slot % nqb;
    } else {
        bh = blockIdx.x / nqb, qblk = blockIdx.x % nqb;
    }
    const int b = bh / H, h = bh % H;
    const int r = lane & 31, hh = lane >> 5;
    const bool active = qblk * QB + wid * 32 < Lq;  // wave-uniform

    // ---- Q fragments (B operand of S^T = K Q^T): element j <-> d = 16kk + 8hh + j
    const long qrow = (long)qblk * QB + wid * 32 + r;
    const bf16_t* qp = Q + (size_t)b * qsb + (size_t)(qrow < Lq ? qrow : Lq - 1) * qsl + (size_t)h * qsh + 8 * hh;
    bf16x8 qf[KK];
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) qf[kk] = *reinterpret_cast<const bf16x8*>(qp + 16 * kk);

    // ---- DMA: lane-constant offsets + the tile's offset, both in the range-checked vector offset
    const int kbytes = (int)(((long)(Lk - 1) * ksl + HD) * 2);  // (checked by the launcher: < 2^31)
    const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(K + (size_t)b * ksb + (size_t)h * ksh), 0, kbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(V + (size_t)b * ksb + (size_t)h * ksh), 0, kbytes, 0x00020000);
    unsigned koff[PPW], voff[PPW];
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        const int row = (wid + NW * i) * G::ROWS_PER_PIECE + lane / G::CHUNKS, pch = lane % G::CHUNKS;
        koff[i] = (unsigned)(row * ksl + G::kswz(row, pch) * 8) * 2u;
        voff[i] = (unsigned)(row * ksl + G::vswz(row, pch) * 8) * 2u;
    }
    const int tile_stride = (int)(KB * ksl * 2);
    auto stage = [&](int tile, char* buf) {
        // The tile offset goes into the VECTOR offset: the hardware range-checks voffset + inst_offset against
        // num_records and leaves the scalar offset out of the check (LLVM AMDGPUUsage, raw buffer intrinsics: "soffset
        // ... excluded from bounds checking"), and the rows past Lk of the last tile MUST read as zeros -- their P is 0,
        // but 0 x NaN from whatever lies behind the last (batch, head) would poison the PV and ones products.
        const unsigned so = (unsigned)(tile * tile_stride);
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rk, (ISP_LDS void*)(buf + (wid + NW * i) * 1024), 16, koff[i] + so, 0, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rv, (ISP_LDS void*)(buf + G::TILE + (wid + NW * i) * 1024), 16, voff[i] + so,
                                                     0, 0, 0);
        }
    };

    // ---- fragment addresses (as in the generic kernel)
    int k_off[2][KK];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
        const int row = kb * 32 + r;
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) k_off[kb][kk] = row * G::ROW + (G::kswz(row, 2 * kk + hh) << 4);
    }
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

    // row sums: a ones-row MFMA per key sub-block at head_dim 64 (one of 9 MFMAs, off the vector pipe), vector adds at
    // head_dim 128 (the 4 extra MFMAs per tile were 2.5 % of the launch there; -DISP_ATT_ONES=0 / 1 forces either)
    constexpr bool ONES = ISP_ATT_ONES < 0 ? HD == 64 : ISP_ATT_ONES != 0;
    f32x16 o[DB], negm;
    [[maybe_unused]] f32x16 lacc;
    constexpr short kOne = F16 ? 0x3c00 : 0x3f80;  // 1.0 in half / bf16
    [[maybe_unused]] const bf16x8 ones = {kOne, kOne, kOne, kOne, kOne, kOne, kOne, kOne};
    [[maybe_unused]] float l_run = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
#pragma unroll
        for (int db = 0; db < DB; ++db) o[db][i] = 0.f;
        negm[i] = 0.f;
        if constexpr (ONES) lacc[i] = 0.f;
    }
    float m_run = 0.f;

    // one key block (32 keys) of a tile: S^T chain from -m, optional mask of keys >= Lk
    auto scores = [&](f32x16& s, const char* buf, int kb, int key0, bool mask) {
        s = negm;
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) {
            const bf16x8 kf = *reinterpret_cast<const bf16x8*>(buf + k_off[kb][kk]);
            s = att_mfma<F16>(kf, qf[kk], s);
        }
        if (mask) {
#pragma unroll
            for (int i = 0; i < 16; ++i)
                if (key0 + (i & 3) + 8 * (i >> 2) + 4 * hh >= Lk) s[i] = -INFINITY;
        }
    };
    auto max16 = [](const f32x16& s) {
        float a = fmaxf(fmaxf(s[0], s[1]), s[2]);
#pragma unroll
        for (int i = 3; i < 15; i += 2) a = fmaxf(fmaxf(a, s[i]), s[i + 1]);
        return fmaxf(a, s[15]);
    };
    // P = exp2(S') in place, then O^T += V^T P^T (and l += 1^T P^T) for one key block
    auto exp_pv = [&](f32x16& s, const char* buf, int kb) {
        float psum = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            s[i] = __builtin_amdgcn_exp2f(PRESCALED ? s[i] : s[i] * c);
            if constexpr (!ONES) psum += s[i];
        }
        if constexpr (!ONES) l_run += psum;
#pragma unroll
        for (int ss = 0; ss < 2; ++ss) {
            bf16x8 pf;
#pragma unroll
            for (int j = 0; j < 8; j += 2) {
                const unsigned pk = pack2o<!F16>(s[8 * ss + j], s[8 * ss + j + 1]);
                pf[j] = (short)(pk & 0xffffu), pf[j + 1] = (short)(pk >> 16);
            }
#pragma unroll
            for (int db = 0; db < DB; ++db) {
                const s16x4 lo = tr_read(buf + v_off[db][kb][ss][0]);
                const s16x4 hi = tr_read(buf + v_off[db][kb][ss][1]);
                const bf16x8 vf = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                o[db] = att_mfma<F16>(vf, pf, o[db]);
            }
            if constexpr (ONES) lacc = att_mfma<F16>(ones, pf, lacc);
        }
    };
    // move the reference maximum by d (per row; the same in both halves of a row) before the tile's exponentials
    auto rebase = [&](f32x16& s0, f32x16& s1, float d, bool first) {
        const float alpha = first ? 1.f : __builtin_amdgcn_exp2f(-d * c);  // (first tile: O = l = 0, and d may be < 0)
        m_run += d;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            s0[i] -= d, s1[i] -= d;
#pragma unroll
            for (int db = 0; db < DB; ++db) o[db][i] *= alpha;
            if constexpr (ONES) lacc[i] *= alpha;
            negm[i] = -m_run;
        }
        if constexpr (!ONES) l_run *= alpha;
    };

    const int nt = (Lk + KB - 1) / KB;
    // FIRST: the reference maximum is set from the tile (whatever its sign); LAST: keys >= Lk are masked and the
    // second key block may be empty.  (Separate instantiations: as run-time flags the masks become 120 selects per tile.)
    auto tile = [&]<bool FIRST, bool LAST>(int t) {
        const char* buf = (t & 1) ? smem + 2 * G::TILE : smem;
        if (!LAST) stage(t + 1, (t & 1) ? smem : smem + 2 * G::TILE);
        if (active) {
            const int key0 = t * KB;
            const bool two = !LAST || key0 + 32 < Lk;  // the second key block holds a key (wave-uniform)
            f32x16 s0, s1;
            scores(s0, buf, 0, key0, LAST);
            if (two) {
                scores(s1, buf, 1, key0 + 32, LAST);
            } else {
#pragma unroll
                for (int i = 0; i < 16; ++i) s1[i] = -INFINITY;
            }
            const float mx = xhalf_max(fmaxf(max16(s0), max16(s1)));  // tile maximum of the row, relative to m_run
            if (FIRST) {
                rebase(s0, s1, mx, true);
            } else if (__builtin_amdgcn_ballot_w64(mx > thr) != 0) {
                rebase(s0, s1, fmaxf(mx, 0.f), false);
            }
            exp_pv(s0, buf, 0);
            if (two) exp_pv(s1, buf, 1);
        }
        __syncthreads();
    };
    stage(0, smem);
    __syncthreads();
    if (nt == 1) {
        tile.template operator()<true, true>(0);
    } else {
        tile.template operator()<true, false>(0);
        for (int t = 1; t + 1 < nt; ++t) tile.template operator()<false, false>(t);
        tile.template operator()<false, true>(nt - 1);
    }
    if (!active) return;

    // ---- epilogue: O[b, q, h, d] = o / l
    float l_tot;
    if constexpr (ONES) l_tot = lacc[0];
    else l_tot = l_run + __shfl_xor(l_run, 32);
    const float inv = 1.0f / l_tot;
    if (lse && hh == 0 && qrow < Lq) lse[(size_t)bh * lse_ld + qrow] = m_run * c + log2f(l_tot);
    if (qrow < Lq) {
        bf16_t* op = O + (size_t)b * osb + (size_t)qrow * osl + (size_t)h * osh;
#pragma unroll
        for (int db = 0; db < DB; ++db)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int d = db * 32 + 8 * g4 + 4 * hh;
                *reinterpret_cast<uint2*>(op + d) =
                    make_uint2(pack2o_sat<!F16>(o[db][4 * g4 + 0] * inv, o[db][4 * g4 + 1] * inv),
                               pack2o_sat<!F16>(o[db][4 * g4 + 2] * inv, o[db][4 * g4 + 3] * inv));
            }
    }
}

// (thin non-template kernels: a __global__ template whose launch bounds depend on a template parameter loses its host-side
// handle under hipcc 7.2 -- the library then fails to load with an undefined symbol)
#define ISP_ATT64_KERNEL(NAME, PRE, F16, HD, NW)                                                                                  \
    __global__ __launch_bounds__(64 * NW, 2) void NAME(const bf16_t* __restrict__ Q, const bf16_t* __restrict__ K,                \
                                                       const bf16_t* __restrict__ V, bf16_t* __restrict__ O, int H, int Lq, int Lk, \
                                                       long qsb, long qsl, long qsh, long ksb, long ksl, long ksh, long osb,       \
                                                       long osl, long osh, float c_arg, float* __restrict__ lse, long lse_ld,     \
                                                       int nbh, int nqb) {                                                        \
        attention64_body<PRE, F16, HD, NW>(Q, K, V, O, H, Lq, Lk, qsb, qsl, qsh, ksb, ksl, ksh, osb, osl, osh, c_arg, lse, lse_ld, \
                                           nbh, nqb);                                                                             \
    }
ISP_ATT64_KERNEL(attention64_kernel_scale_bf16, false, false, 64, 4)
ISP_ATT64_KERNEL(attention64_kernel_bf16, true, false, 64, 4)
ISP_ATT64_KERNEL(attention64_kernel_f16, true, true, 64, 4)
ISP_ATT64_KERNEL(attention128_kernel_bf16, true, false, 128, 8)
ISP_ATT64_KERNEL(attention128_kernel_f16, true, true, 128, 8)
ISP_ATT64_KERNEL(attention128_kernel_bf16_w4, true, false, 128, 4)  // (128-query blocks, two per CU: the default)
ISP_ATT64_KERNEL(attention128_kernel_f16_w4, true, true, 128, 4)
#undef ISP_ATT64_KERNEL

constexpr int kAtt64Lds = Geo<64>::LDS;  // (K + V) x 2 buffers

template <bool PRESCALED, bool F16 = false, int HD = 64, int NW = 4>
int launch_attention64(const void* Q, const void* K, const void* V, void* O, int B, int H, int Lq, int Lk, long qsb,
                       long qsl, long qsh, long ksb, long ksl, long ksh, long osb, long osl, long osh, float scale,
                       float* lse, long lse_ld, hipStream_t s) {
    static bool attr_done = false;
    void (*kern)(const bf16_t*, const bf16_t*, const bf16_t*, bf16_t*, int, int, int, long, long, long, long, long, long, long, long, long,
                 float, float*, long, int, int);
    if constexpr (HD == 128) {
        static_assert(PRESCALED && (NW == 8 || NW == 4), "head_dim 128: base-2-logit queries, 8 or 4 waves");
        if constexpr (NW == 8) kern = F16 ? attention128_kernel_f16 : attention128_kernel_bf16;
        else kern = F16 ? attention128_kernel_f16_w4 : attention128_kernel_bf16_w4;
    } else {
        static_assert(HD == 64 && NW == 4 && (PRESCALED || !F16));
        kern = !PRESCALED ? attention64_kernel_scale_bf16 : (F16 ? attention64_kernel_f16 : attention64_kernel_bf16);
    }
    constexpr int kLds = Geo<HD>::LDS;
    if (!attr_done) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, kLds) != hipSuccess)
            return ISP_ERR_LAUNCH;
        attr_done = true;
    }
    const int nqb = (Lq + 32 * NW - 1) / (32 * NW);
    kern<<<(unsigned)(nqb * B * H), 64 * NW, kLds, s>>>((const bf16_t*)Q, (const bf16_t*)K, (const bf16_t*)V, (bf16_t*)O, H, Lq,
                                                            Lk, qsb, qsl, qsh, ksb, ksl, ksh, osb, osl, osh,
                                                            scale * 1.4426950408889634f, lse, lse_ld, B * H, nqb);
    return isp_launch_status();
}

}  // namespace

// head_dim 128 on the deferred-maximum kernel: 32-bit addressable K / V slice (ISEGPROBE_ATT128_DM=0: the generic kernel)
static bool dm128_ok(int Lk, long kv_stride_l) {
    static const bool off = [] { const char* e = getenv("ISEGPROBE_ATT128_DM"); return e && e[0] == '0'; }();
    return !off && kv_stride_l >= 128 && ((long)(Lk - 1) * kv_stride_l + 128) * 2 < (1L << 31) &&
           (long)KB * kv_stride_l * 2 * ((Lk + KB - 1) / KB) < (1L << 31);
}

extern "C" int isp_attention_pipe_supported(int head_dim, int Lq, int Lk, long kv_stride_l);
extern "C" int isp_attention_fwd_pipe(const void* Q, const void* K, const void* V, void* O, int B, int H, int Lq, int Lk,
                                      int head_dim, long q_stride_b, long q_stride_l, long q_stride_h, long kv_stride_b,
                                      long kv_stride_l, long kv_stride_h, long o_stride_b, long o_stride_l, long o_stride_h,
                                      int f16, void* stream);

// Base-2-logit problems the pipelined kernel takes (csrc/attention_pipe.hip: 256-query workgroups).  A short query
// remainder (the ViT's 1025th token: Lq % 256 == 1) would cost a whole extra round of workgroups there; those queries go
// to the 32-query kernel `rest` instead (one 128-query block per (batch, head), a few microseconds).
template <class Rest>
static int attention_pipe_split(const void* Q, const void* K, const void* V, void* O, int B, int H, int Lq, int Lk, int hd,
                                long qsb, long qsl, long qsh, long ksb, long ksl, long ksh, long osb, long osl, long osh, int f16,
                                void* stream, Rest&& rest) {
    const int rem = Lq % 256;
    const int main_q = (rem != 0 && rem <= 64 && Lq > 256) ? Lq - rem : Lq;
    int rc = isp_attention_fwd_pipe(Q, K, V, O, B, H, main_q, Lk, hd, qsb, qsl, qsh, ksb, ksl, ksh, osb, osl, osh, f16, stream);
    if (rc != ISP_OK || main_q == Lq) return rc;
    return rest((const bf16_t*)Q + (size_t)main_q * qsl, (bf16_t*)O + (size_t)main_q * osl, Lq - main_q);
}

static int attention_fwd_impl(const void* Q, const void* K, const void* V, void* O, int B, int H, int Lq, int Lk,
                              int head_dim, long q_stride_b, long q_stride_l, long q_stride_h, long kv_stride_b,
                              long kv_stride_l, long kv_stride_h, long o_stride_b, long o_stride_l, long o_stride_h,
                              float scale, float* lse, long lse_ld, void* stream, bool logit2 = false) {
    ISP_CHECK_ARG(Q && K && V && O && B > 0 && H > 0 && Lq > 0 && Lk > 0 && (logit2 || scale > 0.f));
    ISP_CHECK_ARG((long)B * H <= 65535);
    // 16-byte vector loads / 8-byte stores need aligned strides
    ISP_CHECK_ARG(q_stride_b % 8 == 0 && q_stride_l % 8 == 0 && q_stride_h % 8 == 0);
    ISP_CHECK_ARG(kv_stride_b % 8 == 0 && kv_stride_l % 8 == 0 && kv_stride_h % 8 == 0);
    ISP_CHECK_ARG(o_stride_b % 4 == 0 && o_stride_l % 4 == 0 && o_stride_h % 4 == 0);
    ISP_CHECK_ARG(!lse || lse_ld >= Lq);
    hipStream_t s = (hipStream_t)stream;
    if (logit2 && !lse && isp_attention_pipe_supported(head_dim, Lq, Lk, kv_stride_l)) {
        auto rest = [&](const bf16_t* q2, bf16_t* o2, int lq2) {
            if (head_dim == 64)
                return launch_attention64<true>(q2, K, V, o2, B, H, lq2, Lk, q_stride_b, q_stride_l, q_stride_h, kv_stride_b, kv_stride_l,
                                                kv_stride_h, o_stride_b, o_stride_l, o_stride_h, 1.f, nullptr, 0, s);
            return launch_attention<128>(q2, K, V, o2, B, H, lq2, Lk, q_stride_b, q_stride_l, q_stride_h, kv_stride_b, kv_stride_l,
                                         kv_stride_h, o_stride_b, o_stride_l, o_stride_h, 0.6931471805599453f, nullptr, 0, s);
        };
        return attention_pipe_split(Q, K, V, O, B, H, Lq, Lk, head_dim, q_stride_b, q_stride_l, q_stride_h, kv_stride_b, kv_stride_l,
                                    kv_stride_h, o_stride_b, o_stride_l, o_stride_h, 0, stream, rest);
    }
    // (generic kernels: their scale argument times log2(e) multiplies the scores; base-2 logits need 1 / log2(e))
    if (logit2 && head_dim != 64) scale = 0.6931471805599453f;
    if (head_dim == 64) {
        // the 64-wide variant addresses keys through a 32-bit buffer range and assumes a key row >= its head slice
        static const bool generic64 = [] { const char* e = getenv("ISEGPROBE_ATT64"); return e && e[0] == '0'; }();
        if (!generic64 && kv_stride_l >= 64 && ((long)(Lk - 1) * kv_stride_l + 64) * 2 < (1L << 31) &&
            (long)KB * kv_stride_l * 2 * ((Lk + KB - 1) / KB) < (1L << 31))
            return logit2 ? launch_attention64<true>(Q, K, V, O, B, H, Lq, Lk, q_stride_b, q_stride_l, q_stride_h, kv_stride_b,
                                                     kv_stride_l, kv_stride_h, o_stride_b, o_stride_l, o_stride_h, 1.f, lse,
                                                     lse_ld, s)
                          : launch_attention64<false>(Q, K, V, O, B, H, Lq, Lk, q_stride_b, q_stride_l, q_stride_h, kv_stride_b,
                                      kv_stride_l, kv_stride_h, o_stride_b, o_stride_l, o_stride_h, scale, lse, lse_ld, s);
        // (generic kernel: its scale argument times log2(e) multiplies the scores; base-2 logits need 1 / log2(e))
        return launch_attention<64>(Q, K, V, O, B, H, Lq, Lk, q_stride_b, q_stride_l, q_stride_h, kv_stride_b, kv_stride_l,
                                    kv_stride_h, o_stride_b, o_stride_l, o_stride_h, logit2 ? 0.6931471805599453f : scale, lse,
                                    lse_ld, s);
    }
    if (head_dim == 128) {
        // 256-query blocks once they still fill the chip several times over
        // 128-query workgroups, two per CU (independent barriers), measured 2 % ahead of one 256-query workgroup per CU although
        // K / V are then streamed twice as often (3.88 vs 3.97 ms per LoftUp launch); ISEGPROBE_ATT128_NW=8: the 8-wave form
        static const bool w4 = [] { const char* e = getenv("ISEGPROBE_ATT128_NW"); return !(e && e[0] == '8'); }();
        if (w4 && logit2 && !lse && (long)((Lq + 255) / 256) * B * H >= 2048 && dm128_ok(Lk, kv_stride_l))
            return launch_attention64<true, false, 128, 4>(Q, K, V, O, B, H, Lq, Lk, q_stride_b, q_stride_l, q_stride_h, kv_stride_b,
                                                           kv_stride_l, kv_stride_h, o_stride_b, o_stride_l, o_stride_h, 1.f, nullptr, 0, s);
        if (logit2 && !lse && (long)((Lq + 255) / 256) * B * H >= 2048 && dm128_ok(Lk, kv_stride_l))
            return launch_attention64<true, false, 128, 8>(Q, K, V, O, B, H, Lq, Lk, q_stride_b, q_stride_l, q_stride_h, kv_stride_b,
                                                           kv_stride_l, kv_stride_h, o_stride_b, o_stride_l, o_stride_h, 1.f, nullptr, 0, s);
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

// Q already carries softmax scale x log2(e) (Q K^T are base-2 logits) -- what the ViT trunk (head_dim 64) and LoftUp's
// inference stream (head_dim 128 / 256) pass
// after folding that factor into the Q rows of its qkv weights (reference dinov2/layers/attention.py:62 applies the
// scale to q after the projection; same product, one rounding).
extern "C" int isp_attention_fwd_logit2(const void* Q, const void* K, const void* V, void* O, int B, int H, int Lq, int Lk,
                                        int head_dim, long q_stride_b, long q_stride_l, long q_stride_h, long kv_stride_b,
                                        long kv_stride_l, long kv_stride_h, long o_stride_b, long o_stride_l,
                                        long o_stride_h, void* stream) {
    return attention_fwd_impl(Q, K, V, O, B, H, Lq, Lk, head_dim, q_stride_b, q_stride_l, q_stride_h, kv_stride_b,
                              kv_stride_l, kv_stride_h, o_stride_b, o_stride_l, o_stride_h, 1.f, nullptr, 0, stream, true);
}

// IEEE-half Q, K, V, O (and probabilities), head_dim 64 / 128 / 256 on the generic kernel: LoftUp's cross-attention in its half-precision
// inference stream (same layouts, strides and kernels as isp_attention_fwd).
extern "C" int isp_attention_fwd_f16(const void* Q, const void* K, const void* V, void* O, int B, int H, int Lq, int Lk,
                                     int head_dim, long q_stride_b, long q_stride_l, long q_stride_h, long kv_stride_b,
                                     long kv_stride_l, long kv_stride_h, long o_stride_b, long o_stride_l, long o_stride_h,
                                     float scale, void* stream) {
    ISP_CHECK_ARG(Q && K && V && O && B > 0 && H > 0 && Lq > 0 && Lk > 0 && scale > 0.f);
    ISP_CHECK_ARG((long)B * H <= 65535);
    ISP_CHECK_ARG(q_stride_b % 8 == 0 && q_stride_l % 8 == 0 && q_stride_h % 8 == 0);
    ISP_CHECK_ARG(kv_stride_b % 8 == 0 && kv_stride_l % 8 == 0 && kv_stride_h % 8 == 0);
    ISP_CHECK_ARG(o_stride_b % 4 == 0 && o_stride_l % 4 == 0 && o_stride_h % 4 == 0);
    hipStream_t s = (hipStream_t)stream;
    const bool wide = (long)((Lq + 255) / 256) * B * H >= 2048;  // 256-query blocks once they still fill the chip several times
#define ISP_ATT_F16(HD, NW)                                                                                                  \
    launch_attention<HD, NW, true>(Q, K, V, O, B, H, Lq, Lk, q_stride_b, q_stride_l, q_stride_h, kv_stride_b, kv_stride_l, \
                                   kv_stride_h, o_stride_b, o_stride_l, o_stride_h, scale, nullptr, 0, s)
    if (head_dim == 64) return ISP_ATT_F16(64, 4);  // (LoftUp on narrow feature maps: head dims <= 64)
    if (head_dim == 128) return wide ? ISP_ATT_F16(128, 8) : ISP_ATT_F16(128, 4);
    if (head_dim == 256) return wide ? ISP_ATT_F16(256, 8) : ISP_ATT_F16(256, 4);
#undef ISP_ATT_F16
    return ISP_ERR_UNSUPPORTED;
}

// isp_attention_fwd_logit2 on IEEE-half Q, K, V, O: the ViT trunk's half-precision inference stream (head_dim 64).
extern "C" int isp_attention_fwd_logit2_f16(const void* Q, const void* K, const void* V, void* O, int B, int H, int Lq, int Lk,
                                            int head_dim, long q_stride_b, long q_stride_l, long q_stride_h,
                                            long kv_stride_b, long kv_stride_l, long kv_stride_h, long o_stride_b,
                                            long o_stride_l, long o_stride_h, void* stream) {
    ISP_CHECK_ARG(Q && K && V && O && B > 0 && H > 0 && Lq > 0 && Lk > 0);
    ISP_CHECK_ARG((long)B * H <= 65535);
    ISP_CHECK_ARG(q_stride_b % 8 == 0 && q_stride_l % 8 == 0 && q_stride_h % 8 == 0);
    ISP_CHECK_ARG(kv_stride_b % 8 == 0 && kv_stride_l % 8 == 0 && kv_stride_h % 8 == 0);
    ISP_CHECK_ARG(o_stride_b % 4 == 0 && o_stride_l % 4 == 0 && o_stride_h % 4 == 0);
    hipStream_t hs = (hipStream_t)stream;
    const bool wide = (long)((Lq + 255) / 256) * B * H >= 2048;
    auto generic = [&](const void* q2, void* o2, int lq2) -> int {  // base-2 logits on the generic kernel: scale = 1 / log2(e)
#define ISP_ATT_F16G(HD, NW)                                                                                                     \
    launch_attention<HD, NW, true>(q2, K, V, o2, B, H, lq2, Lk, q_stride_b, q_stride_l, q_stride_h, kv_stride_b, kv_stride_l, \
                                   kv_stride_h, o_stride_b, o_stride_l, o_stride_h, 0.6931471805599453f, nullptr, 0, hs)
        // 128-query workgroups, two per CU (independent barriers), measured 2 % ahead of one 256-query workgroup per CU although
        // K / V are then streamed twice as often (3.88 vs 3.97 ms per LoftUp launch); ISEGPROBE_ATT128_NW=8: the 8-wave form
        static const bool w4 = [] { const char* e = getenv("ISEGPROBE_ATT128_NW"); return !(e && e[0] == '8'); }();
        if (w4 && head_dim == 128 && wide && dm128_ok(Lk, kv_stride_l))
            return launch_attention64<true, true, 128, 4>(q2, K, V, o2, B, H, lq2, Lk, q_stride_b, q_stride_l, q_stride_h, kv_stride_b,
                                                          kv_stride_l, kv_stride_h, o_stride_b, o_stride_l, o_stride_h, 1.f, nullptr, 0, hs);
        if (head_dim == 128 && wide && dm128_ok(Lk, kv_stride_l))  // deferred-maximum kernel, 8 waves (see attention64_kernel)
            return launch_attention64<true, true, 128, 8>(q2, K, V, o2, B, H, lq2, Lk, q_stride_b, q_stride_l, q_stride_h, kv_stride_b,
                                                          kv_stride_l, kv_stride_h, o_stride_b, o_stride_l, o_stride_h, 1.f, nullptr, 0, hs);
        if (head_dim == 128) return wide ? ISP_ATT_F16G(128, 8) : ISP_ATT_F16G(128, 4);
        if (head_dim == 256) return wide ? ISP_ATT_F16G(256, 8) : ISP_ATT_F16G(256, 4);
        return ISP_ERR_UNSUPPORTED;
#undef ISP_ATT_F16G
    };
    if (isp_attention_pipe_supported(head_dim, Lq, Lk, kv_stride_l)) {
        auto rest = [&](const bf16_t* q2, bf16_t* o2, int lq2) -> int {
            if (head_dim == 64)
                return launch_attention64<true, true>(q2, K, V, o2, B, H, lq2, Lk, q_stride_b, q_stride_l, q_stride_h, kv_stride_b,
                                                      kv_stride_l, kv_stride_h, o_stride_b, o_stride_l, o_stride_h, 1.f, nullptr, 0, hs);
            return generic(q2, o2, lq2);
        };
        return attention_pipe_split(Q, K, V, O, B, H, Lq, Lk, head_dim, q_stride_b, q_stride_l, q_stride_h, kv_stride_b, kv_stride_l,
                                    kv_stride_h, o_stride_b, o_stride_l, o_stride_h, 1, stream, rest);
    }
    if (head_dim != 64) return generic(Q, O, Lq);
    if (!(kv_stride_l >= 64 && ((long)(Lk - 1) * kv_stride_l + 64) * 2 < (1L << 31) &&
          (long)KB * kv_stride_l * 2 * ((Lk + KB - 1) / KB) < (1L << 31)))
        return ISP_ERR_UNSUPPORTED;
    return launch_attention64<true, true>(Q, K, V, O, B, H, Lq, Lk, q_stride_b, q_stride_l, q_stride_h, kv_stride_b, kv_stride_l,
                                          kv_stride_h, o_stride_b, o_stride_l, o_stride_h, 1.f, nullptr, 0, (hipStream_t)stream);
}
