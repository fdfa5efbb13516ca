// Token-stationary fused kernels for the DINOv2 blocks on gfx950.
//
//   isp_vit_mlp_fused:   x += ls2 * fc2(GELU(fc1(LayerNorm(x))))      -- reference dinov2/layers/block.py:92-117 (second
//                         residual branch), mlp.py:34-40, layer_scale.py:25-26, nn.LayerNorm(eps) DINOv2.py:98
//
// Why a different structure from the tile engine (gemm.hip).  With D = 384 the two MLP GEMMs have 6 / 24 K-steps per
// 128 x 128 tile and stage 65 FLOP per byte moved L2 -> LDS: they sit on that staging rate (460-530 TFLOP/s), write and
// re-read the [M, 4D] hidden map through HBM (200 MB per block at batch 32) and pay two prologues / epilogues plus a
// LayerNorm pass.  Here a workgroup OWNS 128 tokens (4 waves x 32, one wave per SIMD, up to 512 registers):
//   * the wave's 32 LayerNorm-ed token rows stay in registers for the whole kernel as MFMA B operands (D/16 k-steps x 4
//     VGPRs = 96 VGPRs at D = 384): the fp32 residual stream is read once, normalised in registers;
//   * only WEIGHTS stream through LDS (one 16 KiB tile = 128 rows x 64 k per 16 MFMAs 32x32x16 per wave, LDS-DMA into a
//     6-deep ring, 5 tiles in flight, one barrier per tile, every fragment read issued 8 MFMAs ahead of its use):
//     131 FLOP per staged byte, the hidden map never leaves registers;
//   * products are formed transposed, H^T = W1 . X^T and Y^T = W2 . H^T, so the 32x32 accumulator layout of H^T (a lane
//     holds 4 x 4 consecutive hidden units of one token) IS, after bias + GELU + bf16 packing, the B operand of the second
//     product -- 8 accumulator registers give the 8 k-slots of a lane, in an order that the host bakes into W2's packed
//     hidden axis (vit_mlp_pack in hip_ops.py); v_mfma_f32_32x32x16 holds the vector issue port for 8 of its 32 cycles
//     (8 of 16 for 16x16x32), which is what lets the GELU's VALU work run beside the MFMAs;
//   * the hidden axis is walked in chunks of 64 and the second product lags the first by one chunk: while the MFMAs of
//     Y^T += W2[:, chunk j-1] . H^T[j-1] run, the VALU turns the accumulators of chunk j into the next B operands;
//   * LayerNorm's affine is folded into fc1 (W1 diag(g), b1 + W1 b) and LayerScale into fc2 at pack time, so the
//     prologue is x_hat = (x - mean) * rstd and the epilogue x += acc + b2.
// Built for D = 384 (DINOv2-S/14): output accumulators D/32 x 16 = 192 registers + 96 operand registers per lane.
#include <type_traits>

#include "isp_common.h"

namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16_t;

constexpr int TILE_BYTES = 16384;  // 128 rows x 64 k bf16, 128-byte rows, 16-B chunks XOR-swizzled by (row >> 1) & 7
constexpr int NSTAGE = 6;          // LDS ring depth; a tile is issued NSTAGE positions ahead, into the slot of the tile
                                   // whose last fragment has just been read
constexpr int TOK_BLOCK = 128;     // tokens per workgroup (4 waves x 32)

template <int V>
using IC = std::integral_constant<int, V>;

typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
__device__ __forceinline__ unsigned cvt_pk_bf16(float lo, float hi) {  // one v_cvt_pk_bf16_f32
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{lo, hi}, bf16x2_t));
}

// 16-bit operand format of the kernel: bf16, or IEEE half (the ViT trunk's default inference stream: three more mantissa bits
// on the normalised rows, the hidden map and the weights at the same MFMA rate; values saturate at half's range)
template <bool F16>
__device__ __forceinline__ f32x16_t mfma32(bf16x8 a, bf16x8 b, f32x16_t c) {
    if constexpr (F16)
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
template <bool F16, bool SAT = true>
__device__ __forceinline__ unsigned pack16(float lo, float hi) {
    if constexpr (F16) return SAT ? pack2h_sat(lo, hi) : pack2h(lo, hi);
    else return cvt_pk_bf16(lo, hi);
}

// Weight tiles, hidden axis in chunks of 64:
//   A-tile (fc1)  (chunk c, kk < KK = D/128): LDS rows 0-63 = hidden 64c + r over k 128kk .. +63, rows 64-127 = the same
//                 hidden units over k 128kk+64 .. +127 -> 2 accumulator tiles (32 hidden x 32 tokens) x 8 k-steps of 16.
//   B-tile (fc2)  (chunk c, nb < NB = D/128): rows = output channels 128nb + r, 64 (permuted) hidden positions of chunk c
//                 -> 4 accumulator tiles x 4 k-steps.
// Stream order per 128-token tile:  A(0) | A(1) B(0) | A(2) B(1) | ... | A(CH-1) B(CH-2) | B(CH-1)   (each group KK + NB = 6
// tiles), continuing seamlessly into the next token tile of the workgroup (or into harmless re-loads at the very end).
template <int D, int HID, bool F16>
__global__ __launch_bounds__(256, 1) void vit_mlp_fused_kernel(float* __restrict__ x, const bf16_t* __restrict__ w1,
                                                               const float* __restrict__ b1, const bf16_t* __restrict__ w2,
                                                               const float* __restrict__ b2, long M, float eps, int n_tiles, int rot_mul,
                                                               int tiles_per_image, int rows_per_image, int first_row, int T) {
    constexpr int KS = D / 16;     // k-steps of 16 over the embedding dim
    constexpr int KK = D / 128;    // A-tiles per chunk
    constexpr int NB = D / 128;    // B-tiles per chunk
    constexpr int NT32 = D / 32;   // output accumulator tiles
    constexpr int CH = HID / 64;   // hidden chunks
    static_assert(D == 384 && HID % 128 == 0 && KK + NB == NSTAGE, "geometry: one A group + one B group per ring turn");
    extern __shared__ __attribute__((aligned(16))) char ring[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 31, lg = lane >> 5;
    // Tiles walk the hidden chunks in rotated order (logical chunk j = physical chunk (j + rot) mod CH; the second product sums
    // over chunks, so any order is the same function up to fp32 summation order): with one order for all, the 256 workgroups
    // ask the L2 for the same 16 KiB weight tile at the same moment (measured: 130 -> 124 us per launch, 123 -> 115 us on the
    // per-image tiling).  The rotation is a function of the tile's position INSIDE its image, so an image's result does not
    // depend on its batch index or batch size; xcd_remap hands each XCD a contiguous run of tiles = all positions.
    auto tile_of = [&](int t) { return xcd_remap(t, n_tiles); };
    auto rot_of = [&](int tile) {
        const int pos = tiles_per_image > 0 ? tile % tiles_per_image : tile;
        return (int)(((unsigned)pos * (unsigned)rot_mul) % (unsigned)CH);
    };
    int rot = rot_of(tile_of(blockIdx.x)), rot_nx = rot;  // current tile's / next tile's (the stream runs across tiles)
    auto phys = [&](int j, int r) {
        const int c = j + r;
        return c >= CH ? c - CH : c;
    };

    // ---- weight stream: buffer_load ... lds against SGPR resources (a 32-bit byte offset per lane and piece, the tile
    // base as a scalar offset: no 64-bit pointers in VGPRs).  Piece p (1 KiB = 8 LDS rows): lane -> (row 8p + lane/8,
    // physical chunk lane%8), source chunk = physical ^ ((row >> 1) & 7); wave `wid` moves pieces wid, wid+4, wid+8, wid+12.
    const __amdgpu_buffer_rsrc_t r_w1 = __builtin_amdgcn_make_buffer_rsrc((void*)w1, 0, HID * D * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_w2 = __builtin_amdgcn_make_buffer_rsrc((void*)w2, 0, HID * D * 2, 0x00020000);
    unsigned offA[4], offB[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = (wid + 4 * i) * 8 + (lane >> 3);
        const int sc = ((lane & 7) ^ ((row >> 1) & 7)) * 8;
        offA[i] = (unsigned)((row & 63) * D + 64 * (row >> 6) + sc) * 2u;
        offB[i] = (unsigned)(row * HID + sc) * 2u;
    }
    auto issue_a = [&](int c, auto kktag, auto slottag, int r) {
        constexpr int kk = decltype(kktag)::value, slot = decltype(slottag)::value;
        const int base = (64 * phys(c, r) * D + 128 * kk) * 2;
#ifdef ISP_ABLATE_MLP_NO_DMA  // timing experiment only: compute on whatever the ring holds
        if (base != -12345) return;
#endif
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r_w1, (ISP_LDS void*)(ring + slot * TILE_BYTES + (wid + 4 * i) * 1024), 16,
                                                     offA[i], base, 0, 0);
    };
    auto issue_b = [&](int c, auto nbtag, auto slottag) {
        constexpr int nb = decltype(nbtag)::value, slot = decltype(slottag)::value;
        const int base = (128 * nb * HID + 64 * phys(c, rot)) * 2;
#ifdef ISP_ABLATE_MLP_NO_DMA
        if (base != -12345) return;
#endif
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r_w2, (ISP_LDS void*)(ring + slot * TILE_BYTES + (wid + 4 * i) * 1024), 16,
                                                     offB[i], base, 0, 0);
    };
    // fragment addresses: 32-row tile rt, row lr, k-step ks of 16 (chunk 2ks + lg); ds_read offsets are 16-bit, the ring is
    // 96 KiB: one address set for slots 0-3 and one (+64 KiB) for slots 4-5, everything else is an immediate
    int loff[2][4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        loff[0][ks] = lr * 128 + (((2 * ks + lg) ^ ((lr >> 1) & 7)) * 16);
        loff[1][ks] = loff[0][ks] + 65536;
    }
    // fragment i (0..15) of the tile in `slot`.  A-tile: MFMA i feeds accumulator tile i/8 with k-step kq = i%8 -> LDS row
    // tile (i/8) + 2*(kq/4), ks = kq%4.  B-tile: output sub-tile i/4, ks = i%4.
#ifdef ISP_ABLATE_MLP_NO_FRAG
    const bf16x8 fconst = __builtin_bit_cast(bf16x8, make_uint4(lane * 0x10001u, 0x3f803f80u, lane, 0x3f803f80u));
#endif
    auto frag = [&](auto slottag, auto atag, auto itag) {
        constexpr int slot = decltype(slottag)::value, I = decltype(itag)::value;
        constexpr bool ATILE = decltype(atag)::value != 0;
        constexpr int rt = ATILE ? (I / 8) + 2 * ((I % 8) / 4) : I / 4;
        constexpr int hi = slot >= 4 ? 1 : 0;
#ifdef ISP_ABLATE_MLP_NO_FRAG  // timing experiment only: no LDS fragment reads
        return fconst;
#else
        return *reinterpret_cast<const bf16x8*>(ring + loff[hi][I % 4] + (slot * TILE_BYTES - hi * 65536 + rt * 4096));
#endif
    };
    // Rolling window of 8 fragments (32 VGPRs): MFMA i of a tile consumes f[i % 8]; the ds_read issued right behind it
    // refills that register with the fragment 8 MFMAs ahead (this tile's second half, then the next tile's first half).
    bf16x8 f[8];

    // Half-way through every tile: wait until the NEXT tile has landed (all but the 4 newer tiles' pieces of this wave),
    // own LDS reads retired (this tile's 16 fragments are all in registers or in flight -> complete), barrier (everyone's
    // pieces landed, everyone is done with this tile's slot), refill this tile's slot with the tile NSTAGE positions on.
    auto sync = [&]() {
        // nothing moves across: left alone the scheduler sinks the 8 MFMAs in front of this point (register-only, free to
        // cross the asm) below it, which leaves their 8 fragment reads bunched up in front of the lgkmcnt(0)
        __builtin_amdgcn_sched_barrier(0);
#ifdef ISP_ABLATE_MLP_NO_DMA
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#else
        asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
#endif
#ifndef ISP_ABLATE_MLP_NO_BARRIER  // timing experiment only (with NO_DMA)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
#endif
        __builtin_amdgcn_sched_barrier(0);
    };

    // prime: stream positions 0..5 = A(0) (slots 0-2), A(1) (slots 3-5); then the first tile's first 8 fragments
    issue_a(0, IC<0>{}, IC<0>{}, rot);
    issue_a(0, IC<1>{}, IC<1>{}, rot);
    issue_a(0, IC<2>{}, IC<2>{}, rot);
    issue_a(1, IC<0>{}, IC<3>{}, rot);
    issue_a(1, IC<1>{}, IC<4>{}, rot);
    issue_a(1, IC<2>{}, IC<5>{}, rot);
#ifdef ISP_ABLATE_MLP_NO_DMA
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#else
    asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
#endif
    __builtin_amdgcn_s_barrier();
    f[0] = frag(IC<0>{}, IC<1>{}, IC<0>{});
    f[1] = frag(IC<0>{}, IC<1>{}, IC<1>{});
    f[2] = frag(IC<0>{}, IC<1>{}, IC<2>{});
    f[3] = frag(IC<0>{}, IC<1>{}, IC<3>{});
    f[4] = frag(IC<0>{}, IC<1>{}, IC<4>{});
    f[5] = frag(IC<0>{}, IC<1>{}, IC<5>{});
    f[6] = frag(IC<0>{}, IC<1>{}, IC<6>{});
    f[7] = frag(IC<0>{}, IC<1>{}, IC<7>{});

    for (int t_lin = blockIdx.x; t_lin < n_tiles; t_lin += gridDim.x) {
        const int tile = tile_of(t_lin);
        rot = rot_nx;
        rot_nx = t_lin + (int)gridDim.x < n_tiles ? rot_of(tile_of(t_lin + gridDim.x)) : 0;
        // rows of a tile: M contiguous rows, or (tiles_per_image > 0) the T patch-token rows first_row .. first_row + T - 1 of
        // each image's rows_per_image rows -- the class-token rows are left to the caller, so that B x 1024 patch tokens
        // are exactly B x 8 tiles (32 x 1025 rows as one range are 256 tiles + 32 rows = a second round of workgroups)
        long tok, tokc;
        bool live;
        if (tiles_per_image > 0) {
            const int img = tile / tiles_per_image;
            const int t = (tile - img * tiles_per_image) * TOK_BLOCK + wid * 32 + lr;
            live = t < T;
            tok = tokc = (long)img * rows_per_image + first_row + (live ? t : T - 1);
        } else {
            tok = (long)tile * TOK_BLOCK + wid * 32 + lr;
            live = tok < M;
            tokc = live ? tok : M - 1;
        }
        // ---- LayerNorm prologue: the wave's 32 rows, normalised, as bf16 B operands (lane: token lr, 8 k of k-step s at 8*lg)
        bf16x8 act[KS];
        {
            const float* row = x + (size_t)tokc * D + 8 * lg;
            float v[KS][8];
            float s1 = 0.f;
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const float4 a = *reinterpret_cast<const float4*>(row + 16 * s);
                const float4 b = *reinterpret_cast<const float4*>(row + 16 * s + 4);
                v[s][0] = a.x, v[s][1] = a.y, v[s][2] = a.z, v[s][3] = a.w;
                v[s][4] = b.x, v[s][5] = b.y, v[s][6] = b.z, v[s][7] = b.w;
#pragma unroll
                for (int e = 0; e < 8; ++e) s1 += v[s][e];
            }
            s1 += __shfl_xor(s1, 32);
            const float mean = s1 * (1.0f / D);
            float s2 = 0.f;
#pragma unroll
            for (int s = 0; s < KS; ++s)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float d = v[s][e] - mean;
                    s2 += d * d;
                }
            s2 += __shfl_xor(s2, 32);
            const float rstd = rsqrtf(s2 * (1.0f / D) + eps);
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                unsigned p[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) p[e] = pack16<F16, false>((v[s][2 * e] - mean) * rstd, (v[s][2 * e + 1] - mean) * rstd);  // |x_hat| <= sqrt(D)
                act[s] = __builtin_bit_cast(bf16x8, make_uint4(p[0], p[1], p[2], p[3]));
            }
        }
        f32x16_t oacc[NT32];
#pragma unroll
        for (int i = 0; i < NT32; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) oacc[i][e] = 0.f;
        f32x16_t hacc[2];
        bf16x8 hb[4];  // the B operands of the chunk whose B-tiles run now

        // bias + GELU of accumulator tile T of chunk c -> hbn[2T], hbn[2T+1].  The biases come by SCALAR loads (wave-uniform
        // address; readfirstlane pins them to SGPRs so the per-lane choice stays a select of two scalars): a
        // register-destination VMEM load inside the DMA stream would make the compiler drain the LDS-DMA pipeline.
        auto gelu_half = [&](int c, auto ttag, auto utag, bf16x8 (&hbn)[4]) {
            constexpr int T = decltype(ttag)::value, u = decltype(utag)::value;
            const float* bb = b1 + 64 * phys(c, rot) + 32 * T + 16 * u;
            float r[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {  // accumulator registers 8u + e = rows 16u + 8*(e/4) + 4*lg + e%4
                const int rowlo = 8 * (e / 4) + (e % 4);
                const float lo = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, bb[rowlo])));
                const float hi = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, bb[rowlo + 4])));
#ifdef ISP_ABLATE_MLP_NO_GELU  // timing experiment only
                r[e] = hacc[T][8 * u + e] + (lg ? hi : lo);
#else
                r[e] = gelu_sig5(hacc[T][8 * u + e] + (lg ? hi : lo));
#endif
            }
            hbn[2 * T + u] = __builtin_bit_cast(bf16x8, make_uint4(pack16<F16>(r[0], r[1]), pack16<F16>(r[2], r[3]),
                                                                 pack16<F16>(r[4], r[5]), pack16<F16>(r[6], r[7])));
        };
        // Pin the instruction order of the 8 MFMAs just emitted: MFMA, its fragment read, then NV VALU instructions (the
        // GELU work that runs beside a B-tile) -- left alone the scheduler clusters the reads and exposes their latency.
        auto pin = [&](auto nvtag) {
            constexpr int NV = decltype(nvtag)::value;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                if constexpr (NV > 0) __builtin_amdgcn_sched_group_barrier(0x002, NV, 0);
            }
        };
        // one A-tile (chunk c, kk) in `slot`: MFMAs 0-7 -> hacc[0], 8-15 -> hacc[1]; `nslot` / NA describe the next tile of
        // the stream (its fragments 0-7 are read behind MFMAs 8-15); refill() issues the tile NSTAGE positions ahead.
        auto a_tile = [&](auto kktag, auto slottag, auto nslottag, auto natag, auto&& refill) {
            constexpr int kk = decltype(kktag)::value;
            using S = decltype(slottag);
            using NS = decltype(nslottag);
            using NA = decltype(natag);
            auto mfma = [&](auto itag) {
                constexpr int I = decltype(itag)::value, T = I / 8;
                if constexpr (kk == 0 && I % 8 == 0) {
                    f32x16_t z;
#pragma unroll
                    for (int e = 0; e < 16; ++e) z[e] = 0.f;
                    hacc[T] = mfma32<F16>(f[I % 8], act[8 * kk + I % 8], z);
                } else {
                    hacc[T] = mfma32<F16>(f[I % 8], act[8 * kk + I % 8], hacc[T]);
                }
                if constexpr (I < 8)
                    f[I % 8] = frag(S{}, IC<1>{}, IC<I + 8>{});
                else
                    f[I % 8] = frag(NS{}, NA{}, IC<I - 8>{});
            };
            mfma(IC<0>{}); mfma(IC<1>{}); mfma(IC<2>{}); mfma(IC<3>{});
            mfma(IC<4>{}); mfma(IC<5>{}); mfma(IC<6>{}); mfma(IC<7>{});
            pin(IC<0>{});
            sync();
            refill();
            mfma(IC<8>{}); mfma(IC<9>{}); mfma(IC<10>{}); mfma(IC<11>{});
            mfma(IC<12>{}); mfma(IC<13>{}); mfma(IC<14>{}); mfma(IC<15>{});
            pin(IC<0>{});
        };
        // one B-tile (output block nb) on the operands hbc; `valu(i)` is called behind MFMA i: GELU work of the next chunk
        auto b_tile = [&](auto nbtag, auto slottag, auto nslottag, auto natag, bf16x8 (&hbc)[4], auto&& refill, auto&& valu,
                          auto vtag) {
            constexpr int nb = decltype(nbtag)::value;
            using S = decltype(slottag);
            using NS = decltype(nslottag);
            using NA = decltype(natag);
            auto mfma = [&](auto itag) {
                constexpr int I = decltype(itag)::value;
                oacc[4 * nb + I / 4] = mfma32<F16>(f[I % 8], hbc[I % 4], oacc[4 * nb + I / 4]);
                if constexpr (I < 8)
                    f[I % 8] = frag(S{}, IC<0>{}, IC<I + 8>{});
                else
                    f[I % 8] = frag(NS{}, NA{}, IC<I - 8>{});
                valu(itag);
            };
            mfma(IC<0>{}); mfma(IC<1>{}); mfma(IC<2>{}); mfma(IC<3>{});
            mfma(IC<4>{}); mfma(IC<5>{}); mfma(IC<6>{}); mfma(IC<7>{});
            pin(vtag);
            sync();
            refill();
            mfma(IC<8>{}); mfma(IC<9>{}); mfma(IC<10>{}); mfma(IC<11>{});
            mfma(IC<12>{}); mfma(IC<13>{}); mfma(IC<14>{}); mfma(IC<15>{});
            pin(vtag);
        };
        auto no_valu = [](auto) {};

#ifndef ISP_ABLATE_MLP_NO_COMPUTE  // timing experiment only: LayerNorm prologue + residual epilogue alone
        // ---- group 0: A(0) in slots 0-2 (6 ahead: B(0) into the same slots); its GELU is the one exposed per token tile
        a_tile(IC<0>{}, IC<0>{}, IC<1>{}, IC<1>{}, [&] { issue_b(0, IC<0>{}, IC<0>{}); });
        a_tile(IC<1>{}, IC<1>{}, IC<2>{}, IC<1>{}, [&] { issue_b(0, IC<1>{}, IC<1>{}); });
        a_tile(IC<2>{}, IC<2>{}, IC<3>{}, IC<1>{}, [&] { issue_b(0, IC<2>{}, IC<2>{}); });
        gelu_half(0, IC<0>{}, IC<0>{}, hb);
        gelu_half(0, IC<0>{}, IC<1>{}, hb);
        gelu_half(0, IC<1>{}, IC<0>{}, hb);
        gelu_half(0, IC<1>{}, IC<1>{}, hb);
        // ---- groups j = 1 .. CH-1: A(j) in slots 3-5, B(j-1) in slots 0-2 with the GELU of chunk j beside its MFMAs.
        // 6 positions ahead of A(j, kk): A(j+1, kk); from the last group: B(CH-1, kk), both into slots 3-5.
        // 6 ahead of B(j-1, nb): B(j, nb); from the last group (B(CH-2)): A(0) of the next token tile, slots 0-2.
        // The last group is peeled (straight-line code after the loop) so that the loop body has a single path.
        auto group = [&](int j, auto lasttag) {
            constexpr bool LAST = decltype(lasttag)::value != 0;
            bf16x8 hbn[4];
            auto gelu_beside = [&](auto ttag) {
                return [&, ttag](auto it) {
                    if constexpr (decltype(it)::value == 0) gelu_half(j, ttag, IC<0>{}, hbn);
                    if constexpr (decltype(it)::value == 8) gelu_half(j, ttag, IC<1>{}, hbn);
                };
            };
            auto refill_a = [&](auto kktag) {
                return [&, kktag] {
                    if constexpr (LAST) issue_b(j, kktag, IC<3 + decltype(kktag)::value>{});
                    else issue_a(j + 1, kktag, IC<3 + decltype(kktag)::value>{}, rot);
                };
            };
            auto refill_b = [&](auto nbtag) {
                return [&, nbtag] {
                    if constexpr (LAST) issue_a(0, nbtag, nbtag, rot_nx);
                    else issue_b(j, nbtag, nbtag);
                };
            };
            a_tile(IC<0>{}, IC<3>{}, IC<4>{}, IC<1>{}, refill_a(IC<0>{}));
            a_tile(IC<1>{}, IC<4>{}, IC<5>{}, IC<1>{}, refill_a(IC<1>{}));
            a_tile(IC<2>{}, IC<5>{}, IC<0>{}, IC<0>{}, refill_a(IC<2>{}));
            b_tile(IC<0>{}, IC<0>{}, IC<1>{}, IC<0>{}, hb, refill_b(IC<0>{}), gelu_beside(IC<0>{}), IC<10>{});
            b_tile(IC<1>{}, IC<1>{}, IC<2>{}, IC<0>{}, hb, refill_b(IC<1>{}), gelu_beside(IC<1>{}), IC<10>{});
            b_tile(IC<2>{}, IC<2>{}, IC<3>{}, IC<(LAST ? 0 : 1)>{}, hb, refill_b(IC<2>{}), no_valu, IC<0>{});
#pragma unroll
            for (int i = 0; i < 4; ++i) hb[i] = hbn[i];
            // keep the group's VALU work (GELU of chunk j) inside the group: moved into the next group's A-tiles it would
            // need copies of both accumulator tiles (the compiler did that: 240 accvgpr moves per group)
            __builtin_amdgcn_sched_barrier(0);
        };
#pragma unroll 1
        for (int j = 1; j + 1 < CH; ++j) group(j, IC<0>{});
        group(CH - 1, IC<1>{});
        // ---- last group: B(CH-1) in slots 3-5 (6 ahead: A(1) of the next token tile, same slots); next tile: A(0) in slot 0
        b_tile(IC<0>{}, IC<3>{}, IC<4>{}, IC<0>{}, hb, [&] { issue_a(1, IC<0>{}, IC<3>{}, rot_nx); }, no_valu, IC<0>{});
        b_tile(IC<1>{}, IC<4>{}, IC<5>{}, IC<0>{}, hb, [&] { issue_a(1, IC<1>{}, IC<4>{}, rot_nx); }, no_valu, IC<0>{});
        b_tile(IC<2>{}, IC<5>{}, IC<0>{}, IC<1>{}, hb, [&] { issue_a(1, IC<2>{}, IC<5>{}, rot_nx); }, no_valu, IC<0>{});

#else
        oacc[0][0] = __builtin_bit_cast(float, (int)act[0][0]) + __builtin_bit_cast(float, (int)act[KS - 1][3]);
#endif
        // ---- epilogue: x += acc + b2 (LayerScale folded into W2 / b2).  A lane owns, per 32-row tile, 4 x 4 consecutive
        // channels (rows 8i + 4*lg .. +3) of token lr.
        if (live) {
            float* row = x + (size_t)tok * D + 4 * lg;
#pragma unroll
            for (int nt = 0; nt < NT32; ++nt)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float4 bb = *reinterpret_cast<const float4*>(b2 + 32 * nt + 8 * i + 4 * lg);
                    float4 r = *reinterpret_cast<const float4*>(row + 32 * nt + 8 * i);
                    r.x += oacc[nt][4 * i + 0] + bb.x;
                    r.y += oacc[nt][4 * i + 1] + bb.y;
                    r.z += oacc[nt][4 * i + 2] + bb.z;
                    r.w += oacc[nt][4 * i + 3] + bb.w;
                    *reinterpret_cast<float4*>(row + 32 * nt + 8 * i) = r;
                }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the ring's last (unused) refills must land before the LDS is released
}

template <int D, int HID, bool F16>
int launch_mlp(float* x, const void* w1, const float* b1, const void* w2, const float* b2, long M, float eps, int images,
               int rows_per_image, int first_row, int T, hipStream_t s) {
    const int tiles_per_image = images > 0 ? (T + TOK_BLOCK - 1) / TOK_BLOCK : 0;
    const int n_tiles = images > 0 ? images * tiles_per_image : (int)((M + TOK_BLOCK - 1) / TOK_BLOCK);
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t p;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess) return ISP_ERR_LAUNCH;
        cus = p.multiProcessorCount;
    }
    const int lds = NSTAGE * TILE_BYTES;
    auto kern = vit_mlp_fused_kernel<D, HID, F16>;
    static bool attr = false;
    if (!attr) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return ISP_ERR_LAUNCH;
        attr = true;
    }
    const int grid = n_tiles < cus ? n_tiles : cus;
    static int rot_mul = -1;
    if (rot_mul < 0) {
        const char* e = getenv("ISEGPROBE_MLP_ROT");
        rot_mul = e ? atoi(e) : 3;  // 8 tiles per image at 448^2 -> chunk offsets 0, 3, .., 21 of 24
    }
    kern<<<grid, 256, lds, s>>>(x, (const bf16_t*)w1, b1, (const bf16_t*)w2, b2, M, eps, n_tiles, rot_mul, tiles_per_image,
                               rows_per_image, first_row, T);
    return isp_launch_status();
}

}  // namespace

extern "C" int isp_vit_mlp_fused(float* x, const void* w1, const float* b1, const void* w2p, const float* b2, long M, int D,
                                 int HID, float eps, void* stream) {
    ISP_CHECK_ARG(x && w1 && b1 && w2p && b2 && M > 0);
    hipStream_t s = (hipStream_t)stream;
    if (D == 384 && HID == 1536) return launch_mlp<384, 1536, false>(x, w1, b1, w2p, b2, M, eps, 0, 0, 0, 0, s);
    return ISP_ERR_UNSUPPORTED;
}

extern "C" int isp_vit_mlp_fused_rows(float* x, const void* w1, const float* b1, const void* w2p, const float* b2, int images,
                                      int rows_per_image, int first_row, int T, int D, int HID, float eps, int w_dtype, void* stream) {
    ISP_CHECK_ARG(x && w1 && b1 && w2p && b2 && images > 0 && T > 0 && first_row >= 0 && first_row + T <= rows_per_image);
    hipStream_t s = (hipStream_t)stream;
    const long M = (long)images * rows_per_image;
    if (D == 384 && HID == 1536 && w_dtype == ISP_BF16)
        return launch_mlp<384, 1536, false>(x, w1, b1, w2p, b2, M, eps, images, rows_per_image, first_row, T, s);
    if (D == 384 && HID == 1536 && w_dtype == ISP_F16)
        return launch_mlp<384, 1536, true>(x, w1, b1, w2p, b2, M, eps, images, rows_per_image, first_row, T, s);
    return ISP_ERR_UNSUPPORTED;
}
