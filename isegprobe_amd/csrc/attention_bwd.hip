// Backward of softmax(Q K^T * scale) V for head_dim 64, 128 and 256 on gfx950 (bf16 MFMA, fp32 accumulate):
// dQ, dK, dV from Q, K, V, O, dO and the forward's log-sum-exp; the [Lq, Lk] probability matrix
// is recomputed tile by tile and never stored.
//
// Needed for the reference's default training mode (feats_injection_mode="before_backbone",
// models/sbd/dinov2/patch-embed_*.py:40): the click patch-embedding receives its gradient through
// every frozen block of the ViT, i.e. through Attention.forward (dinov2/layers/attention.py:54-71),
// which the reference differentiates with autograd (and materialises [B, heads, N, N] twice).
//
// Two launches share one kernel template.  A block of 4 waves OWNS 64 rows of one side and STREAMS
// 64-row tiles of the other side through LDS:
//   * dK/dV launch: owner = 64 keys   (K, V rows in registers), stream = query tiles (Q, dO);
//   * dQ    launch: owner = 64 queries (Q, dO rows in registers), stream = key tiles   (K, V).
// Per tile and wave (16 owner rows o, 64 streamed rows s):
//   X1[s][o] = stream1[s] . own1[o]   (S or S^T)      X2[s][o] = stream2[s] . own2[o]   (dP or dP^T)
//   P = exp2(X1*c - lse[query]);  dS = P * (X2 - delta[query]) * scale        (fp32, in registers)
//   out^T[d][o] += sum_s streamX[s][d] * {P | dS}[s][o]
// The score products are v_mfma_f32_16x16x32_bf16 (A = streamed rows by ds_read_b128, B = owner rows);
// their accumulator layout (4 consecutive s per lane, o = lane&15) IS the B operand layout of
// v_mfma_f32_16x16x16_bf16, whose A operand (streamed tile transposed) comes from
// ds_read_b64_tr_b16: P / dS never touch LDS.  Streamed tiles are staged twice by LDS-DMA, once
// swizzled for the row reads and once for the transposed reads (source-side XOR swizzles).
#include <algorithm>

#include "isp_common.h"

namespace {

constexpr int TB = 64;  // owner rows per block, streamed rows per tile

typedef __attribute__((ext_vector_type(4))) short s16x4;

__device__ __forceinline__ s16x4 tr_read(const char* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((ISP_LDS s16x4*)p);
}
template <int HD>
struct BGeo {
    static constexpr int ROW = HD * 2;            // bytes per row (128, 256 or 512)
    static constexpr int CHUNKS = ROW / 16;       // 16-byte chunks per row
    static constexpr int STR = HD == 256 ? 32 : 64;  // streamed rows per tile (4 images x 2 stages must fit 160 KiB)
    static constexpr int TILE = STR * ROW;        // one streamed tile image
    static constexpr int ROWS_PER_PIECE = 1024 / ROW;
    static constexpr int PIECES = TILE / 1024;    // 1 KiB DMA pieces per image (8 or 16)
    static constexpr int KK = HD / 32;            // k-steps of the score products
    static constexpr int DT = HD / 16;            // 16-wide column tiles of the outputs
    // ds_read_b128 image: 128-B rows share a bank row in pairs, 256-B rows all start on bank 0
    __device__ static __forceinline__ int rswz(int row, int chunk) {
        return HD == 64 ? chunk ^ ((row >> 1) & 7) : chunk ^ (row & 15);
    }
    // tr-read image: a half-wave touches 8 rows x 32 B; spread the 32-B pair index over the rows
    // (256- and 512-byte rows all start on bank 0: the same XOR serves both)
    __device__ static __forceinline__ int tswz(int row, int chunk) {
        return HD == 64 ? chunk ^ (((row >> 1) & 3) << 1) : chunk ^ ((row & 7) << 1);
    }
};

// delta[b,h,q] = sum_d dO[b,q,h,d] * O[b,q,h,d]
template <int HD>
__global__ void attn_delta_kernel(const bf16_t* __restrict__ O, const bf16_t* __restrict__ dO, float* __restrict__ delta,
                                  int H, int Lq, long osb, long osl, long osh, long ld) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= Lq) return;
    const int b = blockIdx.y / H, h = blockIdx.y % H;
    const size_t off = (size_t)b * osb + (size_t)q * osl + (size_t)h * osh;
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < HD / 8; ++i) {
        const bf16x8 o = *reinterpret_cast<const bf16x8*>(O + off + 8 * i);
        const bf16x8 g = *reinterpret_cast<const bf16x8*>(dO + off + 8 * i);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc += bf2f((bf16_t)o[j]) * bf2f((bf16_t)g[j]);
    }
    delta[(size_t)blockIdx.y * ld + q] = acc;
}

struct Side {  // one side of the attention (queries or keys): two row-major [L, 64] bf16 matrices
    const bf16_t* m1;
    const bf16_t* m2;
    long sb1, sl1, sh1, sb2, sl2, sh2;  // element strides (batch, row, head) of m1 / m2
    int L;
};

// OWN_KEYS: owner side = keys (outputs dK = out1 with m-stream Q, dV = out2 with stream dO);
// otherwise owner side = queries (output dQ = out1 with stream K).
// OW: 16-row owner sub-tiles per wave (the block owns 64*OW rows).  The streamed fragments -- the whole LDS read
// volume -- are shared by the OW sub-tiles: with OW = 1 the head_dim-128 launches read 256 LDS cycles per 1024 MFMA
// cycles per wave, i.e. the LDS port is as busy as the MFMA pipe (330 TFLOP/s on LoftUp's dK/dV); OW = 2 halves that.
template <int HD, bool OWN_KEYS, int OW>
__global__ __launch_bounds__(256) void attn_bwd_kernel(Side own, Side str, const float* __restrict__ lse,
                                                       const float* __restrict__ delta, long ld_stat,
                                                       bf16_t* __restrict__ out1, bf16_t* __restrict__ out2, long ob,
                                                       long ol, long oh, int H, float scale, float c,
                                                       float* __restrict__ part1, float* __restrict__ part2) {
    // part1 / part2 (nullable): split launch -- gridDim.z blocks share the streamed range of one owner block and add
    // their partial results into fp32 [B, own.L, H, HD] buffers (converted by attn_convert_kernel afterwards).
    // LDS per stage: [stream1 rows][stream2 rows][stream1 tr]([stream2 tr] when OWN_KEYS)
    using G = BGeo<HD>;
    constexpr int TILE = G::TILE, KK = G::KK, DT = G::DT;
    constexpr int NT_IMG = OWN_KEYS ? 4 : 3;
    constexpr int STAGE = NT_IMG * TILE;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fq = lane >> 4;
    const int b = blockIdx.y / H, h = blockIdx.y % H;

    // ---- owner rows: B operands of the score products, element j <-> d = 32kk + 8fq + j
    int orow[OW];
    bf16x8 own1[OW][KK], own2[OW][KK];
    float lse_o[OW], delta_o[OW];
    const size_t stat_row = (size_t)blockIdx.y * ld_stat;
#pragma unroll
    for (int ow = 0; ow < OW; ++ow) {
        orow[ow] = (blockIdx.x * OW + ow) * TB + wid * 16 + fr;
        const int oc = orow[ow] < own.L ? orow[ow] : own.L - 1;
        const bf16_t* o1 = own.m1 + (size_t)b * own.sb1 + (size_t)oc * own.sl1 + (size_t)h * own.sh1 + 8 * fq;
        const bf16_t* o2 = own.m2 + (size_t)b * own.sb2 + (size_t)oc * own.sl2 + (size_t)h * own.sh2 + 8 * fq;
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) {
            own1[ow][kk] = *reinterpret_cast<const bf16x8*>(o1 + 32 * kk);
            own2[ow][kk] = *reinterpret_cast<const bf16x8*>(o2 + 32 * kk);
        }
        lse_o[ow] = delta_o[ow] = 0.f;
        if (!OWN_KEYS) {
            lse_o[ow] = lse[stat_row + oc];
            delta_o[ow] = delta[stat_row + oc];
        }
    }

    // ---- DMA: a tile image = PIECES pieces of 1 KiB (1024/ROW rows each); wave w takes pieces w, w+4, ...
    const bf16_t* s1 = str.m1 + (size_t)b * str.sb1 + (size_t)h * str.sh1;
    const bf16_t* s2 = str.m2 + (size_t)b * str.sb2 + (size_t)h * str.sh2;
    auto stage = [&](int tile, char* buf) {
#pragma unroll
        for (int i = 0; i < G::PIECES / 4; ++i) {
            const int piece = wid + 4 * i;
            const int row = piece * G::ROWS_PER_PIECE + lane / G::CHUNKS, pch = lane % G::CHUNKS;
            int g = tile * G::STR + row;
            g = g < str.L ? g : str.L - 1;
            const bf16_t* r1 = s1 + (size_t)g * str.sl1;
            const bf16_t* r2 = s2 + (size_t)g * str.sl2;
            glds16(r1 + G::rswz(row, pch) * 8, buf + piece * 1024);
            glds16(r2 + G::rswz(row, pch) * 8, buf + TILE + piece * 1024);
            glds16(r1 + G::tswz(row, pch) * 8, buf + 2 * TILE + piece * 1024);
            if (OWN_KEYS) glds16(r2 + G::tswz(row, pch) * 8, buf + 3 * TILE + piece * 1024);
        }
    };

    // ---- fragment offsets.  Row image: row 16mi + fr, logical chunk 4kk + fq (swizzle term depends on fr only).
    int r_off[KK];
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) r_off[kk] = fr * G::ROW + (G::rswz(fr, 4 * kk + fq) << 4);
    // Transposed image: lane i = 4q+p of 16-lane group fq addresses row 16mi + 4fq + q, columns 16dt + 4p .. +3;
    // the hardware hands lane fr column 16dt + fr of rows 4fq .. 4fq+3 (A operand of the 16x16x16 product).
    int t_off[DT];
    {
        const int row = 4 * fq + (fr >> 2), colb = 8 * (fr & 3);  // byte offset of the 4 columns within 32 B
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
            t_off[dt] = row * G::ROW + (G::tswz(row, 2 * dt + (colb >> 4)) << 4) + (colb & 15);
    }

    f32x4 acc1[OW][DT], acc2[OW][DT];  // out^T tiles: [dt] -> rows d = 16dt + 4fq + r, column o = fr
#pragma unroll
    for (int ow = 0; ow < OW; ++ow)
#pragma unroll
        for (int i = 0; i < DT; ++i) acc1[ow][i] = acc2[ow][i] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nt_all = (str.L + G::STR - 1) / G::STR;
    const int per = (nt_all + gridDim.z - 1) / gridDim.z;
    const int t_begin = blockIdx.z * per, nt = min(nt_all, t_begin + per);
    if (t_begin >= nt) return;
    // dK/dV launch: log-sum-exp and delta of the streamed (query) rows, one tile ahead in registers.  (Loaded inside the
    // sub-tile loop they were two global round trips per 16 streamed rows in front of the exponentials -- behind the next
    // tile's LDS-DMA in the vmcnt queue, so the first of them also waited for that.)
    constexpr int NMI = G::STR / 16;
    float4 l_nxt[NMI], d_nxt[NMI];
    auto fetch_stats = [&](int t) {
#pragma unroll
        for (int mi = 0; mi < NMI; ++mi) {
            const size_t at = stat_row + (size_t)t * G::STR + 16 * mi + 4 * fq;
            l_nxt[mi] = *reinterpret_cast<const float4*>(lse + at);
            d_nxt[mi] = *reinterpret_cast<const float4*>(delta + at);
        }
    };
    if (OWN_KEYS) fetch_stats(t_begin);
    stage(t_begin, smem);
    __syncthreads();
    for (int t = t_begin; t < nt; ++t) {
        const char* buf = smem + ((t - t_begin) & 1) * STAGE;
        float4 l_cur[NMI], d_cur[NMI];
        if (OWN_KEYS) {
#pragma unroll
            for (int mi = 0; mi < NMI; ++mi) l_cur[mi] = l_nxt[mi], d_cur[mi] = d_nxt[mi];
            if (t + 1 < nt) fetch_stats(t + 1);  // (in front of the DMA issue: older in the vmcnt queue)
        }
        if (t + 1 < nt) stage(t + 1, smem + ((t - t_begin + 1) & 1) * STAGE);
#pragma unroll
        for (int mi = 0; mi < G::STR / 16; ++mi) {  // 16 streamed rows s = 16mi + 4fq + r at a time, owner o = fr
            f32x4 x1[OW], x2[OW];
#pragma unroll
            for (int ow = 0; ow < OW; ++ow) x1[ow] = x2[ow] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < KK; ++kk) {
                const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(buf + mi * 16 * G::ROW + r_off[kk]);
                const bf16x8 a2 = *reinterpret_cast<const bf16x8*>(buf + TILE + mi * 16 * G::ROW + r_off[kk]);
#pragma unroll
                for (int ow = 0; ow < OW; ++ow) {
                    x1[ow] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, own1[ow][kk], x1[ow], 0, 0, 0);
                    x2[ow] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2, own2[ow][kk], x2[ow], 0, 0, 0);
                }
            }
            // P and dS in place (x1 -> P, x2 -> dS)
            const int sbase = t * G::STR + 16 * mi + 4 * fq;
            float l4[4], d4[4];
            if (OWN_KEYS) {  // statistics belong to the streamed (query) rows; buffers are padded to 64
                const float4 lv = l_cur[mi], dv = d_cur[mi];
                l4[0] = lv.x, l4[1] = lv.y, l4[2] = lv.z, l4[3] = lv.w;
                d4[0] = dv.x, d4[1] = dv.y, d4[2] = dv.z, d4[3] = dv.w;
            }
            s16x4 ds[OW], pp[OW];
#pragma unroll
            for (int ow = 0; ow < OW; ++ow) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const bool ok = sbase + r < str.L && orow[ow] < own.L;
                    const float lv = OWN_KEYS ? l4[r] : lse_o[ow], dv = OWN_KEYS ? d4[r] : delta_o[ow];
                    const float p = ok ? __builtin_amdgcn_exp2f(fmaf(x1[ow][r], c, -lv)) : 0.f;
                    pp[ow][r] = (short)f2bf(p);
                    ds[ow][r] = (short)f2bf(ok ? p * (x2[ow][r] - dv) * scale : 0.f);
                }
            }
            // out1^T += stream1^T . dS ; out2^T += stream2^T . P   (contraction over these 16 streamed rows)
            // (all transposed fragments of the sub-tile first: read one by one in front of their MFMAs, every product waited for
            //  its own LDS round trip)
            s16x4 a1v[DT], a2v[DT];
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                a1v[dt] = tr_read(buf + 2 * TILE + mi * 16 * G::ROW + t_off[dt]);
                if (OWN_KEYS) a2v[dt] = tr_read(buf + 3 * TILE + mi * 16 * G::ROW + t_off[dt]);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
#pragma unroll
                for (int ow = 0; ow < OW; ++ow)
                    acc1[ow][dt] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a1v[dt], ds[ow], acc1[ow][dt], 0, 0, 0);
                if (OWN_KEYS) {
#pragma unroll
                    for (int ow = 0; ow < OW; ++ow)
                        acc2[ow][dt] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a2v[dt], pp[ow], acc2[ow][dt], 0, 0, 0);
                }
            }
        }
        __syncthreads();
    }

    if (part1) {  // split launch: fp32 partial sums
#pragma unroll
        for (int ow = 0; ow < OW; ++ow) {
            if (orow[ow] >= own.L) continue;
            const size_t off = (((size_t)b * own.L + orow[ow]) * H + h) * HD + 4 * fq;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    atomicAdd(part1 + off + 16 * dt + r, acc1[ow][dt][r]);
                    if (OWN_KEYS) atomicAdd(part2 + off + 16 * dt + r, acc2[ow][dt][r]);
                }
        }
        return;
    }
#pragma unroll
    for (int ow = 0; ow < OW; ++ow) {
        if (orow[ow] >= own.L) continue;
        const size_t off = (size_t)b * ob + (size_t)orow[ow] * ol + (size_t)h * oh + 4 * fq;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            *reinterpret_cast<uint2*>(out1 + off + 16 * dt) =
                make_uint2(pack2bf(acc1[ow][dt][0], acc1[ow][dt][1]), pack2bf(acc1[ow][dt][2], acc1[ow][dt][3]));
            if (OWN_KEYS)
                *reinterpret_cast<uint2*>(out2 + off + 16 * dt) =
                    make_uint2(pack2bf(acc2[ow][dt][0], acc2[ow][dt][1]), pack2bf(acc2[ow][dt][2], acc2[ow][dt][3]));
        }
    }
}

// fp32 partials [B, L, H, HD] -> bf16 output with the caller's strides
__global__ void attn_convert_kernel(const float* __restrict__ part, bf16_t* __restrict__ out, long ob, long ol, long oh,
                                    int L, int H, int hd, long total4) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total4) return;
    const long e = i * 4;
    const int d = (int)(e % hd);
    long t = e / hd;
    const int h = (int)(t % H);
    t /= H;
    const int l = (int)(t % L);
    const long b = t / L;
    const float4 v = *reinterpret_cast<const float4*>(part + e);
    *reinterpret_cast<uint2*>(out + b * ob + (long)l * ol + (long)h * oh + d) = make_uint2(pack2bf(v.x, v.y), pack2bf(v.z, v.w));
}

template <int HD, bool OWN_KEYS>
int launch_bwd(const Side& own, const Side& str, const float* lse, const float* delta, long ld, void* out1, void* out2,
               long ob, long ol, long oh, int B, int H, float scale, hipStream_t s, float* part = nullptr) {
    // head_dim 128 runs one block per CU anyway (LDS): give each wave 32 owner rows there when the owner side is long
    // enough to still fill the chip; head_dim 64 keeps 16 (2 blocks per CU, more blocks for the short ViT sequences)
    constexpr int lds = 2 * (OWN_KEYS ? 4 : 3) * BGeo<HD>::TILE;
    const bool wide = HD == 128 && (long)((own.L + 2 * TB - 1) / (2 * TB)) * B * H >= 256;
    auto launch = [&](auto kern, int ow) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
            return (int)ISP_ERR_LAUNCH;
        dim3 grid((own.L + ow * TB - 1) / (ow * TB), B * H);
        // Few owner rows and a long streamed side (LoftUp's dK/dV at the 224^2 training crop: 128 blocks for 256 CUs,
        // each sweeping 784 query tiles): split the streamed range over gridDim.z blocks that add fp32 partials.
        const long blocks = (long)grid.x * grid.y;
        const int nt = (str.L + BGeo<HD>::STR - 1) / BGeo<HD>::STR;
        int nsplit = 1;
        if (part && blocks < 256 && nt >= 64) nsplit = (int)std::min<long>((512 + blocks - 1) / blocks, nt / 16);
        if (nsplit > 1) {
            const size_t bytes = (size_t)B * own.L * H * HD * 4;
            float* p1 = part;
            float* p2 = OWN_KEYS ? part + (size_t)B * own.L * H * HD : nullptr;
            if (hipMemsetAsync(part, 0, bytes * (OWN_KEYS ? 2 : 1), s) != hipSuccess) return (int)ISP_ERR_LAUNCH;
            grid.z = nsplit;
            kern<<<grid, 256, lds, s>>>(own, str, lse, delta, ld, (bf16_t*)out1, (bf16_t*)out2, ob, ol, oh, H, scale,
                                       scale * 1.4426950408889634f, p1, p2);
            const long total4 = (long)B * own.L * H * HD / 4;
            attn_convert_kernel<<<(unsigned)((total4 + 255) / 256), 256, 0, s>>>(p1, (bf16_t*)out1, ob, ol, oh, own.L, H, HD, total4);
            if (OWN_KEYS)
                attn_convert_kernel<<<(unsigned)((total4 + 255) / 256), 256, 0, s>>>(p2, (bf16_t*)out2, ob, ol, oh, own.L, H, HD, total4);
            return isp_launch_status();
        }
        kern<<<grid, 256, lds, s>>>(own, str, lse, delta, ld, (bf16_t*)out1, (bf16_t*)out2, ob, ol, oh, H, scale,
                                   scale * 1.4426950408889634f, nullptr, nullptr);
        return isp_launch_status();
    };
    if constexpr (HD == 128) {
        if (wide) return launch(attn_bwd_kernel<HD, OWN_KEYS, 2>, 2);
    }
    return launch(attn_bwd_kernel<HD, OWN_KEYS, 1>, 1);
}

}  // namespace

template <int HD>
int attention_bwd_impl(const void* Q, const void* K, const void* V, const void* O, const void* dO, const float* lse,
                       float* delta, long stat_ld, void* dQ, void* dK, void* dV, int B, int H, int Lq, int Lk, long q_stride_b,
                       long q_stride_l, long q_stride_h, long kv_stride_b, long kv_stride_l, long kv_stride_h,
                       long o_stride_b, long o_stride_l, long o_stride_h, float scale, float* kv_part, hipStream_t s) {
    attn_delta_kernel<HD><<<dim3((Lq + 255) / 256, B * H), 256, 0, s>>>((const bf16_t*)O, (const bf16_t*)dO, delta, H, Lq,
                                                                        o_stride_b, o_stride_l, o_stride_h, stat_ld);
    if (int rc = isp_launch_status()) return rc;
    const Side qs{(const bf16_t*)Q, (const bf16_t*)dO, q_stride_b, q_stride_l, q_stride_h, o_stride_b, o_stride_l, o_stride_h, Lq};
    const Side ks{(const bf16_t*)K, (const bf16_t*)V, kv_stride_b, kv_stride_l, kv_stride_h, kv_stride_b, kv_stride_l, kv_stride_h, Lk};
    // dK (stream1 = Q with dS) and dV (stream2 = dO with P): gradients share the K/V strides
    if (int rc = launch_bwd<HD, true>(ks, qs, lse, delta, stat_ld, dK, dV, kv_stride_b, kv_stride_l, kv_stride_h, B, H, scale, s, kv_part))
        return rc;
    if (!dQ) return ISP_OK;  // caller does not need the query gradient (LoftUp's first layer: queries come from the image)
    // dQ (stream1 = K with dS)
    return launch_bwd<HD, false>(qs, ks, lse, delta, stat_ld, dQ, nullptr, q_stride_b, q_stride_l, q_stride_h, B, H, scale, s);
}

extern "C" int isp_attention_bwd(const void* Q, const void* K, const void* V, const void* O, const void* dO,
                                 const float* lse, float* delta, long stat_ld, void* dQ, void* dK, void* dV, int B, int H,
                                 int Lq, int Lk, int head_dim, long q_stride_b, long q_stride_l, long q_stride_h,
                                 long kv_stride_b, long kv_stride_l, long kv_stride_h, long o_stride_b, long o_stride_l,
                                 long o_stride_h, float scale, float* kv_split_workspace, void* stream) {
    ISP_CHECK_ARG(Q && K && V && O && dO && lse && delta && dK && dV);
    ISP_CHECK_ARG(B > 0 && H > 0 && Lq > 0 && Lk > 0 && scale > 0.f && (long)B * H <= 65535);
    ISP_CHECK_ARG(stat_ld % TB == 0 && stat_ld >= Lq);  // statistics rows padded: float4 reads of a partial last tile
    ISP_CHECK_ARG(q_stride_b % 8 == 0 && q_stride_l % 8 == 0 && q_stride_h % 8 == 0);
    ISP_CHECK_ARG(kv_stride_b % 8 == 0 && kv_stride_l % 8 == 0 && kv_stride_h % 8 == 0);
    ISP_CHECK_ARG(o_stride_b % 8 == 0 && o_stride_l % 8 == 0 && o_stride_h % 8 == 0);
    hipStream_t s = (hipStream_t)stream;
    if (head_dim == 64)
        return attention_bwd_impl<64>(Q, K, V, O, dO, lse, delta, stat_ld, dQ, dK, dV, B, H, Lq, Lk, q_stride_b, q_stride_l,
                                      q_stride_h, kv_stride_b, kv_stride_l, kv_stride_h, o_stride_b, o_stride_l, o_stride_h,
                                      scale, kv_split_workspace, s);
    if (head_dim == 128)
        return attention_bwd_impl<128>(Q, K, V, O, dO, lse, delta, stat_ld, dQ, dK, dV, B, H, Lq, Lk, q_stride_b, q_stride_l,
                                       q_stride_h, kv_stride_b, kv_stride_l, kv_stride_h, o_stride_b, o_stride_l, o_stride_h,
                                       scale, kv_split_workspace, s);
    if (head_dim == 256)
        return attention_bwd_impl<256>(Q, K, V, O, dO, lse, delta, stat_ld, dQ, dK, dV, B, H, Lq, Lk, q_stride_b, q_stride_l,
                                       q_stride_h, kv_stride_b, kv_stride_l, kv_stride_h, o_stride_b, o_stride_l, o_stride_h,
                                       scale, kv_split_workspace, s);
    return ISP_ERR_UNSUPPORTED;
}
