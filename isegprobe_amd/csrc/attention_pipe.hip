// Software-pipelined fused attention forward for gfx950: one wave per SIMD, 64 queries per wave, head_dim 64 / 128.
//
// Replaces, for inference, the forward of Attention.forward (reference dinov2/layers/attention.py:54-71; head_dim 64,
// the ViT trunk) and of nn.MultiheadAttention inside LoftUp's CrossAttentionLayer (reference
// core/model/upsamplers/loftup/layers.py:182-198; head_dim 101 zero-padded to 128).  Same operands, layouts and
// fragment geometry as csrc/attention.hip (swapped product S^T = K Q^T, accumulator-as-operand P, V^T through
// ds_read_b64_tr_b16); what changes is the schedule:
//
//   * a workgroup is 4 waves = one per SIMD, each wave owns 64 queries as TWO 32-query streams that share every K and V
//     fragment read: half the LDS traffic per FLOP of the 32-query kernels (those spend as long on fragment reads as on
//     MFMAs at head_dim 64: DESIGN.md section 4, "52 us skeleton");
//   * with one wave per SIMD nothing else hides the softmax, so the tile loop is a two-stage pipeline in program order:
//       phase A   MFMA: S(t+1) = K(t+1) Q^T for both streams        VALU: P(t) = exp2(S(t)), packed to 16 bit
//       phase B   MFMA: O += V(t)^T P(t)^T for both streams         VALU: row sums of P(t), row maxima of S(t+1)
//     (two named score sets, swapped by unrolling the loop twice); a phase is written step by step -- one fragment
//     read two steps ahead, two MFMAs, a fixed share of the vector work -- and each step is fenced with sched_barrier(0):
//     the MFMAs are asm statements (see mfma_s / mfma_o) and the scheduler cannot interleave around what it cannot see;
//   * the score chains start from -m (the row's reference maximum), so the MFMA output goes straight into v_exp; m moves
//     only when a tile exceeds it by more than 2^ATT_THR (wave-uniform branch at the END of phase B, after the tile's
//     PV product: O, l, S(t+1) and the -m registers are corrected together -- cdna guide T13's safe order);
//   * K / V tiles (64 keys) arrive by buffer_load ... lds into a 3-slot ring, tile t+2 issued at the top of iteration t:
//     one s_waitcnt vmcnt(0) + one s_barrier per tile, both a whole iteration behind their DMA.  The tile offset sits in
//     the range-checked VECTOR offset: rows past the last key read as zeros.
// Q must carry softmax scale x log2(e) (base-2 logits): the callers fold it into the query projection.
// Inference only (no log-sum-exp output); Lk >= 128 (two tiles); the launcher sends everything else to attention.hip.
#include "isp_common.h"

namespace {

constexpr int KB = 64;  // keys per tile
#ifndef ISP_ATT_THR
#define ISP_ATT_THR 6.0f
#endif

typedef __attribute__((ext_vector_type(4))) short s16x4;

__device__ __forceinline__ s16x4 tr_read(const char* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((ISP_LDS s16x4*)p);
}

template <int HD>
struct GeoP {  // LDS image of one 64-key K (or V) tile: as Geo<HD> in attention.hip
    static constexpr int ROW = HD * 2;
    static constexpr int CHUNKS = ROW / 16;
    static constexpr int TILE = KB * ROW;
    static constexpr int ROWS_PER_PIECE = 1024 / ROW;
    static constexpr int PIECES = TILE / 1024;
    static constexpr int SLOT = 2 * TILE;   // K then V
    static constexpr int NSLOT = 3;
    static constexpr int LDS = NSLOT * SLOT;  // 48 KiB (head_dim 64) / 96 KiB (128)
    __device__ static __forceinline__ int kswz(int row, int chunk) {
        return HD == 64 ? chunk ^ ((row >> 1) & 7) : chunk ^ (row & 15);
    }
    __device__ static __forceinline__ int vswz(int row, int chunk) {
        return HD == 64 ? chunk ^ (((row >> 1) & 1) << 2) : chunk ^ ((row & 3) << 2);
    }
};

__device__ __forceinline__ float xhalf_max(float x) {  // max(x of lane, x of lane ^ 32)
    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
}

// ---- MFMAs as asm statements.  With builtins hipcc picks ONE register file for every MFMA destination of a function
// (all VGPR-form, or all AGPR-form as soon as anything names an AGPR), and this kernel needs both: the O accumulators
// and the Q operands (only the matrix pipe touches them) in the accumulator file, the score chains and -m (read by the
// vector unit) in the vector file -- 400+ live registers, of which at most 256 may be VGPRs.  As builtins the score sets
// were parked in AGPRs and came back through 1100 v_accvgpr moves per tile pair.  An asm statement is opaque to the
// scheduler, so the instruction order is the SOURCE order, fenced step by step with sched_barrier(0).  Hazards hipcc
// does not pad for an asm statement: a VALU-written operand needs two states before the MFMA that reads it (the
// s_nop 1 in front of every MFMA: it issues while the previous MFMA occupies the pipe), an MFMA result twelve states
// before any other reader (the s_nop 11 statements at the phase boundaries).
template <bool F16>
__device__ __forceinline__ void mfma_s_init(f32x16& d, const bf16x8& a, const bf16x8& b, const f32x16& c) {  // d = a b + c
    if constexpr (F16) asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, %3" : "=&v"(d) : "v"(a), "a"(b), "v"(c));
    else asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %3" : "=&v"(d) : "v"(a), "a"(b), "v"(c));
}
template <bool F16>
__device__ __forceinline__ void mfma_s(f32x16& d, const bf16x8& a, const bf16x8& b) {  // d += a b (vector file)
    if constexpr (F16) asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(d) : "v"(a), "a"(b));
    else asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(d) : "v"(a), "a"(b));
}
template <bool F16>
__device__ __forceinline__ void mfma_o(f32x16& d, const bf16x8& a, const bf16x8& b) {  // d += a b (accumulator file)
    if constexpr (F16) asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(d) : "v"(a), "v"(b));
    else asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(d) : "v"(a), "v"(b));
}

typedef __attribute__((ext_vector_type(2))) float f32x2p;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2p;
template <bool F16>
__device__ __forceinline__ unsigned pack_p(float lo, float hi) {  // one v_cvt_pk_{bf16,f16}_f32 (P <= 2^ATT_THR: no saturation needed)
    if constexpr (F16) return pack2h(lo, hi);
    else return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2p{lo, hi}, bf16x2p));
}

#define ISP_FENCE() __builtin_amdgcn_sched_barrier(0)
// timing experiments only (tools/att_pipe_variants.sh): ISP_PIPE_ABL is a bit mask -- 1 no exp/pack, 2 no max/sum, 4 no tile DMA
// after the prologue, 8 no PV MFMAs, 16 no QK MFMAs, 32 no per-tile barrier + wait, 64 no fragment reads (stale registers)
#ifndef ISP_PIPE_ABL
#define ISP_PIPE_ABL 0
#endif

template <int HD, bool F16>
__device__ __forceinline__ void attention_pipe_body(const bf16_t* __restrict__ Q, const bf16_t* __restrict__ K,
                                                    const bf16_t* __restrict__ V, bf16_t* __restrict__ O, int H, int Lq, int Lk,
                                                    long qsb, long qsl, long qsh, long ksb, long ksl, long ksh, long osb,
                                                    long osl, long osh, int nbh, int nqb) {
    using G = GeoP<HD>;
    constexpr int KK = HD / 16, DB = HD / 32, NW = 4, QB = 64 * NW, PPW = G::PIECES / NW;
    constexpr float thr = ISP_ATT_THR;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    // q-blocks of one (batch, head) share blockIdx % 8 (= an XCD under round-robin dispatch: its K / V stay in ONE L2)
    int bh, qblk;
    if (nbh % 8 == 0) {
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        bh = (slot / nqb) * 8 + xcd, qblk = slot % nqb;
    } else {
        bh = blockIdx.x / nqb, qblk = blockIdx.x % nqb;
    }
    const int b = bh / H, h = bh % H;
    const int r = lane & 31, hh = lane >> 5;

    // ---- Q fragments of both streams (B operand of S^T = K Q^T): element j <-> d = 16kk + 8hh + j; kept in AGPRs
    long qrow[2];
    bf16x8 qf[2][KK];
#pragma unroll
    for (int st = 0; st < 2; ++st) {
        qrow[st] = (long)qblk * QB + wid * 64 + st * 32 + r;
        const bf16_t* qp = Q + (size_t)b * qsb + (size_t)(qrow[st] < Lq ? qrow[st] : Lq - 1) * qsl + (size_t)h * qsh + 8 * hh;
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) {
            qf[st][kk] = *reinterpret_cast<const bf16x8*>(qp + 16 * kk);
            asm volatile("" : "+a"(qf[st][kk]));
        }
    }

    // ---- DMA: one K and one V piece (1 KiB) per wave and slot index i; offsets in the range-checked vector offset
    const int kbytes = (int)(((long)(Lk - 1) * ksl + HD) * 2);  // (checked by the launcher: < 2^31)
    const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc((void*)(K + (size_t)b * ksb + (size_t)h * ksh), 0, kbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc((void*)(V + (size_t)b * ksb + (size_t)h * ksh), 0, kbytes, 0x00020000);
    // Lane-constant byte offsets: DMA sources (koff / voff: row of the piece, swizzled 16-byte chunk) and fragment
    // addresses relative to a ring slot -- K row kb*32 + r, logical chunk 2kk + hh (the swizzle term depends on r only:
    // kb adds a constant); V^T by transposed reads of 4 consecutive keys x 64 bytes per half wave (attention.hip): key
    // 4hh + gq, column block db; (kb, ss, jj) add (kb*32 + 16ss + 8jj) * ROW (swizzle unchanged: multiples of 8 keys).
    // At head_dim 128 they are RECOMPUTED every iteration from an opaque copy of the lane id: as loop invariants they
    // were the values the allocator spilled to scratch (39 registers), and every reload's s_waitcnt vmcnt(0) drains the
    // tile DMA that was just issued.  At head_dim 64 they fit and stay in registers.
    struct Offs {
        unsigned koff[PPW], voff[PPW];
        int k_off[KK], v_off[DB];
    };
    constexpr bool RECOMPUTE = HD == 128;
    const unsigned tile_stride = (unsigned)(KB * ksl * 2);
    const int ksl_i = (int)ksl;
    auto make_offs = [&](unsigned ln, bool with_dma = true) {  // (unsigned: the divisions by CHUNKS are shifts)
        Offs f;
        const int lr = ln & 31, lh = ln >> 5;
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            if (!with_dma) {
                f.koff[i] = f.voff[i] = 0;
                continue;
            }
            const int row = (wid + NW * i) * G::ROWS_PER_PIECE + (int)(ln / G::CHUNKS), pch = (int)(ln % G::CHUNKS);
            f.koff[i] = (unsigned)(row * ksl_i + G::kswz(row, pch) * 8) * 2u;
            f.voff[i] = (unsigned)(row * ksl_i + G::vswz(row, pch) * 8) * 2u;
        }
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) f.k_off[kk] = lr * G::ROW + (G::kswz(lr, 2 * kk + lh) << 4);
        const int gi = ln & 15, gq = gi >> 2, gp = gi & 3, g1 = (ln >> 4) & 1;
#pragma unroll
        for (int db = 0; db < DB; ++db) {
            const int key = 4 * lh + gq, col = db * 32 + 16 * g1 + 4 * gp;
            f.v_off[db] = G::TILE + key * G::ROW + G::vswz(key, col >> 3) * 16 + (col & 7) * 2;
        }
        return f;
    };
    const Offs offs0 = make_offs((unsigned)lane);
    auto cur_offs = [&]() {
        if constexpr (RECOMPUTE) {
            unsigned ln = (unsigned)lane;
            asm volatile("" : "+v"(ln));
            return make_offs(ln, false);  // (DMA offsets: derived at each piece, see stage_piece)
        } else {
            return offs0;
        }
    };
    // one DMA instruction of tile `tile` into ring slot `slot`: piece index i of the wave, K (which = 0) or V (1)
    auto stage_piece = [&](const Offs& f, int tile, int slot, int i, int which) {
        char* buf = smem + slot * G::SLOT + (which ? G::TILE : 0) + (wid + NW * i) * 1024;
        const unsigned so = (unsigned)tile * tile_stride;
        unsigned off;
        if constexpr (RECOMPUTE) {  // (derived here, from an opaque lane id: as values kept across the phases they were spilled)
            unsigned ln = (unsigned)lane;
            asm volatile("" : "+v"(ln));
            const int row = (wid + NW * i) * G::ROWS_PER_PIECE + (int)(ln / G::CHUNKS), pch = (int)(ln % G::CHUNKS);
            off = (unsigned)(row * ksl_i + (which ? G::vswz(row, pch) : G::kswz(row, pch)) * 8) * 2u;
        } else {
            off = which ? f.voff[i] : f.koff[i];
        }
        __builtin_amdgcn_raw_ptr_buffer_load_lds(which ? rv : rk, (ISP_LDS void*)buf, 16, off + so, 0, 0, 0);
    };
    auto stage = [&](const Offs& f, int tile, int slot) {
        char* buf = smem + slot * G::SLOT;
        const unsigned so = (unsigned)tile * tile_stride;
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rk, (ISP_LDS void*)(buf + (wid + NW * i) * 1024), 16, f.koff[i] + so, 0, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rv, (ISP_LDS void*)(buf + G::TILE + (wid + NW * i) * 1024), 16, f.voff[i] + so, 0, 0, 0);
        }
    };

    f32x16 o[2][DB], negm[2];
    float m_run[2] = {0.f, 0.f}, l_acc[2][2] = {{0.f, 0.f}, {0.f, 0.f}};  // row sums in two chains per stream
#pragma unroll
    for (int st = 0; st < 2; ++st) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            negm[st][i] = 0.f;
#pragma unroll
            for (int db = 0; db < DB; ++db) o[st][db][i] = 0.f;
        }
        asm volatile("" : "+v"(negm[st]));
#pragma unroll
        for (int db = 0; db < DB; ++db) asm volatile("" : "+a"(o[st][db]));
    }

    // S(t) of both streams: [stream][key block]; two named sets alternate between "current" and "next"
    struct Scores {
        f32x16 v[2][2];
    };
    Scores sA, sB;
    unsigned pfu[2][2][2][4];  // packed P(t): [stream][key block][k-slot of 16 keys][dword]

    auto k_frag = [&](const Offs& f, const char* kbuf, int j) {  // step j of the score product: key block j / KK, k-step j % KK
        return *reinterpret_cast<const bf16x8*>(kbuf + f.k_off[j % KK] + (j / KK) * 32 * G::ROW);
    };
    auto v_frag = [&](const Offs& f, const char* vbuf, int j) {  // step j of the PV product, order (kb, ss, db)
        const int db = j % DB, ks = j / DB;                        // ks = 2kb + ss: keys 16 ks ..
        const char* p = vbuf + f.v_off[db] + 16 * ks * G::ROW;
        const s16x4 lo = tr_read(p);
        const s16x4 hi = tr_read(p + 8 * G::ROW);
        return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    };
    auto mask_tail = [&](Scores& sn, int key0) {  // keys >= Lk of the last tile
#pragma unroll
        for (int st = 0; st < 2; ++st)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    if (key0 + kb * 32 + (i & 3) + 8 * (i >> 2) + 4 * hh >= Lk) sn.v[st][kb][i] = -INFINITY;
    };
    // move stream st's reference maximum by d before the exponentials of the tile in `sn` (T13): everything that is
    // relative to it moves together (rare: the O accumulators take a round trip through the vector file)
    auto rebase = [&](Scores& sn, int st, float d, bool first) {
        const float alpha = first ? 1.f : __builtin_amdgcn_exp2f(-d);
        m_run[st] += d;
        l_acc[st][0] *= alpha, l_acc[st][1] *= alpha;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            sn.v[st][0][i] -= d, sn.v[st][1][i] -= d;
            if (!first) {
#pragma unroll
                for (int db = 0; db < DB; ++db) o[st][db][i] *= alpha;
            }
            negm[st][i] = -m_run[st];
        }
        // opaque: seen as sixteen copies of one value the compiler would rebuild the vector in front of every chain
        asm volatile("" : "+v"(negm[st]));
    };
    auto row_max = [&](const Scores& s, int st) {
        float a = -INFINITY;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int i = 0; i < 16; i += 2) a = fmaxf(fmaxf(a, s.v[st][kb][i]), s.v[st][kb][i + 1]);
        return xhalf_max(a);
    };

    auto o_settle = [&]() {
        if constexpr (DB == 2)
            asm volatile("s_nop 11" : "+a"(o[0][0]), "+a"(o[0][1]), "+a"(o[1][0]), "+a"(o[1][1]));
        else
            asm volatile("s_nop 11" : "+a"(o[0][0]), "+a"(o[0][1]), "+a"(o[0][2]), "+a"(o[0][3]), "+a"(o[1][0]), "+a"(o[1][1]), "+a"(o[1][2]),
                         "+a"(o[1][3]));
        ISP_FENCE();
    };
    const int nt = (Lk + KB - 1) / KB;
    const bool ragged = (Lk % KB) != 0;
    // timing experiments only (ISP_PIPE_ABL & 128): shader-clock stamps of block 0 / wave 0, written over the first bytes of O
    unsigned long stamp[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    auto tick = [&](int i) {
        if constexpr ((ISP_PIPE_ABL & 128) != 0) {
            ISP_FENCE();
            stamp[i] = __builtin_amdgcn_s_memtime();
            ISP_FENCE();
        }
    };
    auto tick_add = [&](int i, unsigned long t0) {
        if constexpr ((ISP_PIPE_ABL & 128) != 0) {
            ISP_FENCE();
            stamp[i] += __builtin_amdgcn_s_memtime() - t0;
            ISP_FENCE();
        }
    };
    auto now = [&]() -> unsigned long {
        if constexpr ((ISP_PIPE_ABL & 128) != 0) {
            ISP_FENCE();
            const unsigned long v = __builtin_amdgcn_s_memtime();
            ISP_FENCE();
            return v;
        }
        return 0;
    };
    tick(0);
    constexpr int NA = 2 * KK;        // steps of phase A (one K fragment, two MFMAs each)
    constexpr int NB = 4 * DB;        // steps of phase B (one V fragment, two MFMAs each)
    constexpr int EA = 64 / NA;       // exponentials per phase-A step (over both streams)
    constexpr int EB = 64 / NB;       // summed / compared values per phase-B step
    // Row sums of P(t): beside the PV product (phase B) at head_dim 64, where phase A's vector work (twice the
    // exponentials per MFMA) already outlasts its MFMAs; right behind the exponentials (phase A) at head_dim 128, where
    // that phase has the room and keeping the fp32 P values alive through phase B costs 64 registers the kernel lacks.
    constexpr bool SUM_IN_A = HD == 128;
#ifndef ISP_PIPE_PF
#define ISP_PIPE_PF 2
#endif
    constexpr int PF = ISP_PIPE_PF;                                  // fragment reads run PF steps (2 PF MFMAs) ahead of their use
    constexpr int DMA_EVERY_A = 2 * NA / PPW, DMA_EVERY_B = 2 * NB / PPW;  // slices per DMA instruction (K in phase A, V in B)
    static_assert(DMA_EVERY_A >= 2 && DMA_EVERY_B >= 2, "one DMA instruction per slice at most");

    // exponentials of flat elements [e0, e1) of `sc` (e = 32 st + 16 kb + i), in place
    auto exp_range = [&](Scores& sc, int e0, int e1) {
#pragma unroll
        for (int e = e0; e < e1; ++e) {
            const float p = __builtin_amdgcn_exp2f(sc.v[e >> 5][(e >> 4) & 1][e & 15]);
            sc.v[e >> 5][(e >> 4) & 1][e & 15] = p;
            if constexpr (SUM_IN_A) l_acc[e >> 5][e & 1] += p;
        }
    };
    // 16-bit packing of the value pairs [p0, p1) (pair p = elements 2p, 2p+1): dword (p & 3) of k-slot (p >> 2) & 1
    auto pack_range = [&](const Scores& sc, int p0, int p1) {
#pragma unroll
        for (int p = p0; p < p1; ++p) {
            const f32x16& x = sc.v[p >> 4][(p >> 3) & 1];
            pfu[p >> 4][(p >> 3) & 1][(p >> 2) & 1][p & 3] = pack_p<F16>(x[2 * (p & 7)], x[2 * (p & 7) + 1]);
        }
    };
    auto pfrag = [&](int st, int ks) {  // B operand of the PV product for k-slot ks = 2kb + ss
        return __builtin_bit_cast(bf16x8, make_uint4(pfu[st][ks >> 1][ks & 1][0], pfu[st][ks >> 1][ks & 1][1], pfu[st][ks >> 1][ks & 1][2],
                                                     pfu[st][ks >> 1][ks & 1][3]));
    };

    // ---- phase A of iteration t: S'(t+1) chains from tile t+1's K || exp2 + pack of S'(t).
    // A wave issues in order and the matrix pipe takes one MFMA per 32 cycles: vector work belongs in the ~24 issue cycles
    // BEHIND each MFMA (an MFMA holds the vector issue port for 8 of its 32 cycles), one slice per MFMA -- a first version
    // with the whole step's vector work in one lump in front of an MFMA pair left the pipe idle during the lump and the wave
    // stalled between the pair (111 us against 76 us for the 32-query kernel on the ViT shape).
    auto phase_a = [&](const Offs& f, Scores& sc, Scores& sn, const char* kbuf, bool overlap, int dma_tile) {
        bf16x8 kf[PF + 1];
#pragma unroll
        for (int j = 0; j < PF; ++j) kf[j] = k_frag(f, kbuf, j);
        ISP_FENCE();
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            const int kb = j / KK, kk = j % KK;
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                if (!(ISP_PIPE_ABL & 16)) {
                    if (kk == 0) mfma_s_init<F16>(sn.v[st][kb], kf[j % (PF + 1)], qf[st][kk], negm[st]);
                    else mfma_s<F16>(sn.v[st][kb], kf[j % (PF + 1)], qf[st][kk]);
                }
                ISP_FENCE();
                // the slice behind this MFMA: a fragment read PF steps ahead, half of the step's exponentials, the packs
                // of the previous slice's values (their v_exp results have had a whole MFMA to arrive), and -- one per
                // DMA_EVERY slices -- one DMA instruction of tile t+2's K (its ~100 issue cycles then sit in an MFMA's
                // shadow instead of in front of the phase: 8 of them at the top of an iteration cost ~1000 cycles)
                const int h = 2 * j + st;                      // slice index, 0 .. 2 NA - 1
                if (st == 0 && j + PF < NA && !(ISP_PIPE_ABL & 64)) kf[(j + PF) % (PF + 1)] = k_frag(f, kbuf, j + PF);
                if (dma_tile >= 0 && h % DMA_EVERY_A == 1 && h / DMA_EVERY_A < PPW) stage_piece(f, dma_tile, dma_tile % 3, h / DMA_EVERY_A, 0);
                if (overlap && !(ISP_PIPE_ABL & 1)) {
                    exp_range(sc, h * EA / 2, (h + 1) * EA / 2);
                    if (h > 0) pack_range(sc, (h - 1) * EA / 4, h * EA / 4);
                }
                ISP_FENCE();
            }
        }
        if (overlap && !(ISP_PIPE_ABL & 1)) pack_range(sc, (2 * NA - 1) * EA / 4, 2 * NA * EA / 4);
        // the chains' results are read by the vector unit next (row maxima, masks): twelve states behind the last MFMA
        asm volatile("s_nop 11" : "+v"(sn.v[0][0]), "+v"(sn.v[0][1]), "+v"(sn.v[1][0]), "+v"(sn.v[1][1]));
        ISP_FENCE();
    };
    // ---- phase B of iteration t: O += V(t)^T P(t)^T || row sums of P(t) (`sc`, fp32) and row maxima of S'(t+1) (`sn`)
    auto phase_b = [&](const Offs& f, const Scores& sc, const Scores* sn, const char* vbuf, float (&mx)[2], int dma_tile) {
        float ma[2] = {-INFINITY, -INFINITY};
        bf16x8 vf[PF + 1];
#pragma unroll
        for (int j = 0; j < PF; ++j) vf[j] = v_frag(f, vbuf, j);
        ISP_FENCE();
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int db = j % DB, ks = j / DB;
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                if (!(ISP_PIPE_ABL & 8)) mfma_o<F16>(o[st][db], vf[j % (PF + 1)], pfrag(st, ks));
                ISP_FENCE();
                const int h = 2 * j + st;
                if (st == 0 && j + PF < NB && !(ISP_PIPE_ABL & 64)) vf[(j + PF) % (PF + 1)] = v_frag(f, vbuf, j + PF);
                if (dma_tile >= 0 && h % DMA_EVERY_B == 1 && h / DMA_EVERY_B < PPW) stage_piece(f, dma_tile, dma_tile % 3, h / DMA_EVERY_B, 1);
#pragma unroll
                for (int e = h * EB / 2; e < (h + 1) * EB / 2 && !(ISP_PIPE_ABL & 2); e += 2) {  // element pairs (e, e + 1) of block (e >> 5, (e >> 4) & 1)
                    const int s2 = e >> 5, kb = (e >> 4) & 1, i = e & 15;
                    if constexpr (!SUM_IN_A) {
                        l_acc[s2][0] += sc.v[s2][kb][i];
                        l_acc[s2][1] += sc.v[s2][kb][i + 1];
                    }
                    if (sn) ma[s2] = fmaxf(fmaxf(ma[s2], sn->v[s2][kb][i]), sn->v[s2][kb][i + 1]);
                }
                // opaque per slice: otherwise the two sum chains are re-vectorised across slices into lumps of dependent
                // v_pk_add_f32 (one lump of seven, each behind an s_nop, per four MFMAs)
                // (head_dim 64 only: at 128 the sums sit behind the exponentials, and the extra constraints cost 45 spilled registers)
                if constexpr (!SUM_IN_A)
                    asm volatile("" : "+v"(l_acc[0][0]), "+v"(l_acc[0][1]), "+v"(l_acc[1][0]), "+v"(l_acc[1][1]), "+v"(ma[0]), "+v"(ma[1]));
                ISP_FENCE();
            }
        }
        if (sn) mx[0] = xhalf_max(ma[0]), mx[1] = xhalf_max(ma[1]);
        ISP_FENCE();
        // Twelve states between the last PV MFMA and ANY other reader of the O accumulators -- stated here, with the
        // accumulators as operands, not at the later use: where two code paths join (the loop's back edge, the two tails)
        // the allocator places accumulator copies at the END of the predecessor blocks, i.e. in front of a wait that sits
        // behind the join.  (Found as: stream 1, head-dim block 3 -- the last MFMA of the phase -- short of its last two
        // k-slots on the odd-tile-count path, all-ones V giving 0.78 instead of 1.)
        o_settle();
    };

    // One pipelined iteration: on entry `sc` = S'(t) (row maxima already handled), tile t+1's K and tile t's V resident.
    auto iter = [&](Scores& sc, Scores& sn, int t) {
        // tile t+1 has landed (its DMA was issued a whole iteration ago); every wave is past iteration t-1, whose V tile
        // shares the slot tile t+2 goes to
        const Offs f = cur_offs();
        if (!(ISP_PIPE_ABL & 32)) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
#ifdef ISP_PIPE_DMA_TOP
        if (t + 2 < nt && !(ISP_PIPE_ABL & 4)) stage(f, t + 2, (t + 2) % 3);
        const int dma_tile = -1;
#else
        const int dma_tile = (t + 2 < nt && !(ISP_PIPE_ABL & 4)) ? t + 2 : -1;  // issued piece by piece inside the phases
#endif
        const char* kbuf = smem + ((t + 1) % 3) * G::SLOT;
        const char* vbuf = smem + (t % 3) * G::SLOT;
        ISP_FENCE();
        const unsigned long ta = now();
        phase_a(f, sc, sn, kbuf, true, dma_tile);
        tick_add(5, ta);
        if (ragged && t + 2 == nt) mask_tail(sn, (t + 1) * KB);  // wave-uniform
        ISP_FENCE();
        float mx[2];
        const unsigned long tb = now();
        phase_b(f, sc, &sn, vbuf, mx, dma_tile);
        tick_add(6, tb);
        if (__builtin_amdgcn_ballot_w64(mx[0] > thr || mx[1] > thr) != 0) {  // rare after the first tiles
            rebase(sn, 0, fmaxf(mx[0], 0.f), false);
            rebase(sn, 1, fmaxf(mx[1], 0.f), false);
#pragma unroll
            for (int st = 0; st < 2; ++st)
#pragma unroll
                for (int db = 0; db < DB; ++db) asm volatile("" : "+a"(o[st][db]));
        }
        ISP_FENCE();
    };

    // ---- prologue: tiles 0 and 1 in flight, S'(0) with its own maxima as the first reference
    stage(offs0, 0, 0);
    stage(offs0, 1, 1);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PPW) : "memory");  // tile 0 (the older 2*PPW pieces) has landed
    __builtin_amdgcn_s_barrier();
    ISP_FENCE();
    tick(1);
    phase_a(offs0, sB, sA, smem, false, -1);  // (sB unused)
    rebase(sA, 0, row_max(sA, 0), true);
    rebase(sA, 1, row_max(sA, 1), true);
    ISP_FENCE();

    tick(2);
    // ---- tile loop, unrolled twice for the two score sets: iteration t computes S(t+1) and finishes tile t
    int t = 0;
#pragma unroll 1
    for (; t + 2 <= nt - 1; t += 2) {
        iter(sA, sB, t);
        iter(sB, sA, t + 1);
    }
    // ---- last tile (nt - 1): exponentials and PV product without a successor
    auto tail = [&](Scores& sc, int tl) {
        exp_range(sc, 0, 64);
        pack_range(sc, 0, 32);
        ISP_FENCE();
        float mx[2];
        phase_b(cur_offs(), sc, nullptr, smem + (tl % 3) * G::SLOT, mx, -1);
    };
    tick(3);
    if (t < nt - 1) {
        iter(sA, sB, t);
        tail(sB, nt - 1);
    } else {
        tail(sA, nt - 1);
    }
    tick(4);

    // ---- epilogue: O[b, q, h, d] = o / l
#pragma unroll
    for (int st = 0; st < 2; ++st) {
        const float lsum = l_acc[st][0] + l_acc[st][1];
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(lsum), __float_as_uint(lsum), false, false);
        const float inv = 1.0f / (__uint_as_float(sw[0]) + __uint_as_float(sw[1]));
        if (qrow[st] < Lq) {
            bf16_t* op = O + (size_t)b * osb + (size_t)qrow[st] * osl + (size_t)h * osh;
#pragma unroll
            for (int db = 0; db < DB; ++db)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const int d = db * 32 + 8 * g4 + 4 * hh;
                    *reinterpret_cast<uint2*>(op + d) =
                        make_uint2(pack2o_sat<!F16>(o[st][db][4 * g4 + 0] * inv, o[st][db][4 * g4 + 1] * inv),
                                   pack2o_sat<!F16>(o[st][db][4 * g4 + 2] * inv, o[st][db][4 * g4 + 3] * inv));
                }
        }
    }
    if constexpr ((ISP_PIPE_ABL & 128) != 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        tick(7);
        if (blockIdx.x == 0 && tid == 0) {
            unsigned long* dbg = reinterpret_cast<unsigned long*>(O);
            for (int i = 0; i < 8; ++i) dbg[i] = stamp[i];
        }
    }
}

// (thin non-template kernels: with the body written directly as a __global__ template hipcc 7.2 leaves the kernel's
// host-side handle undefined when the body contains asm statements -- the library then fails to load)
#define ISP_PIPE_KERNEL(NAME, HD, F16)                                                                                          \
    __global__ __launch_bounds__(256, 1) void NAME(const bf16_t* __restrict__ Q, const bf16_t* __restrict__ K,                 \
                                                   const bf16_t* __restrict__ V, bf16_t* __restrict__ O, int H, int Lq, int Lk, \
                                                   long qsb, long qsl, long qsh, long ksb, long ksl, long ksh, long osb,        \
                                                   long osl, long osh, int nbh, int nqb) {                                      \
        attention_pipe_body<HD, F16>(Q, K, V, O, H, Lq, Lk, qsb, qsl, qsh, ksb, ksl, ksh, osb, osl, osh, nbh, nqb);             \
    }
ISP_PIPE_KERNEL(attention_pipe_kernel_64_bf16, 64, false)
ISP_PIPE_KERNEL(attention_pipe_kernel_64_f16, 64, true)
ISP_PIPE_KERNEL(attention_pipe_kernel_128_bf16, 128, false)
ISP_PIPE_KERNEL(attention_pipe_kernel_128_f16, 128, true)
#undef ISP_PIPE_KERNEL

template <int HD, bool F16>
int launch_pipe(const void* Q, const void* K, const void* V, void* O, int B, int H, int Lq, int Lk, long qsb, long qsl,
                long qsh, long ksb, long ksl, long ksh, long osb, long osl, long osh, hipStream_t s) {
    static bool attr_done = false;
    void (*kern)(const bf16_t*, const bf16_t*, const bf16_t*, bf16_t*, int, int, int, long, long, long, long, long, long, long, long, long, int, int);
    if constexpr (HD == 64) kern = F16 ? attention_pipe_kernel_64_f16 : attention_pipe_kernel_64_bf16;
    else kern = F16 ? attention_pipe_kernel_128_f16 : attention_pipe_kernel_128_bf16;
    if (!attr_done) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, GeoP<HD>::LDS) != hipSuccess)
            return ISP_ERR_LAUNCH;
        attr_done = true;
    }
    const int nqb = (Lq + 255) / 256;
    kern<<<(unsigned)(nqb * B * H), 256, GeoP<HD>::LDS, s>>>((const bf16_t*)Q, (const bf16_t*)K, (const bf16_t*)V, (bf16_t*)O, H, Lq, Lk,
                                                              qsb, qsl, qsh, ksb, ksl, ksh, osb, osl, osh, B * H, nqb);
    return isp_launch_status();
}

}  // namespace

static bool pipe_shape_ok(int head_dim, int Lq, int Lk, long kv_stride_l) {
    if (head_dim != 64 && head_dim != 128) return false;
    if (Lk < 2 * KB || Lq < 1 || kv_stride_l < head_dim) return false;
    return ((long)(Lk - 1) * kv_stride_l + head_dim) * 2 < (1L << 31) && (long)KB * kv_stride_l * 2 * ((Lk + KB - 1) / KB) < (1L << 31);
}

// Whether the pipelined kernel takes a problem (the callers in attention.hip fall back to the 32-query kernels otherwise):
// base-2-logit queries, no log-sum-exp output, at least two key tiles, 32-bit addressable K / V slices.
extern "C" int isp_attention_pipe_supported(int head_dim, int Lq, int Lk, long kv_stride_l) {
    // OFF by default (ISEGPROBE_ATT_PIPE=1 turns it on): measured on MI355X it does not beat the 32-query kernels yet --
    // 91 us against 76 us on the ViT shape (B 32 x 6 heads x 1025, head_dim 64), 3.9 ms against 4.0 - 4.2 ms on LoftUp's
    // (B 8 x 4 heads, 200 704 x 1024, head_dim 128).  In-kernel stamps (tools/att_pipe_stamps.py): the MFMAs cost the wave
    // 50 - 70 cycles each instead of 32 and nothing hides behind them; removing ALL vector work leaves their cost unchanged
    // (ablation table in DESIGN.md section 4).  The callers therefore route here only when asked to.
    static const bool on = [] { const char* e = getenv("ISEGPROBE_ATT_PIPE"); return e && e[0] == '1'; }();
    if (!on || (head_dim != 64 && head_dim != 128)) return 0;
    return pipe_shape_ok(head_dim, Lq, Lk, kv_stride_l) ? 1 : 0;
}

// Q [B, Lq, H, hd] (strides in elements), already multiplied by softmax scale x log2(e); K, V [B, Lk, H, hd]; O like Q.
// Queries [0, Lq): the caller splits off a short remainder (Lq % 256) for the 32-query kernels when that saves a round.
extern "C" int isp_attention_fwd_pipe(const void* Q, const void* K, const void* V, void* O, int B, int H, int Lq, int Lk,
                                      int head_dim, long q_stride_b, long q_stride_l, long q_stride_h, long kv_stride_b,
                                      long kv_stride_l, long kv_stride_h, long o_stride_b, long o_stride_l, long o_stride_h,
                                      int f16, void* stream) {
    ISP_CHECK_ARG(Q && K && V && O && B > 0 && H > 0 && Lq > 0 && Lk > 0);
    ISP_CHECK_ARG(q_stride_b % 8 == 0 && q_stride_l % 8 == 0 && q_stride_h % 8 == 0);
    ISP_CHECK_ARG(kv_stride_b % 8 == 0 && kv_stride_l % 8 == 0 && kv_stride_h % 8 == 0);
    ISP_CHECK_ARG(o_stride_b % 4 == 0 && o_stride_l % 4 == 0 && o_stride_h % 4 == 0);
    if (!pipe_shape_ok(head_dim, Lq, Lk, kv_stride_l)) return ISP_ERR_UNSUPPORTED;  // (callable whatever the switch says: tests, benches)
    hipStream_t s = (hipStream_t)stream;
#define ISP_PIPE(HD, F)                                                                                                       \
    launch_pipe<HD, F>(Q, K, V, O, B, H, Lq, Lk, q_stride_b, q_stride_l, q_stride_h, kv_stride_b, kv_stride_l, kv_stride_h, \
                       o_stride_b, o_stride_l, o_stride_h, s)
    if (head_dim == 64) return f16 ? ISP_PIPE(64, true) : ISP_PIPE(64, false);
    return f16 ? ISP_PIPE(128, true) : ISP_PIPE(128, false);
#undef ISP_PIPE
}
