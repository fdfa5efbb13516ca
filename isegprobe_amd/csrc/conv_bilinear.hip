// First head convolution taken THROUGH the bilinear resize (gfx950).
//
//   reference: F.interpolate(x, size=(H, W), mode="bilinear", align_corners=True)   core/model/iseg_probe_model.py:120-129
//              (or the bilinear upsampler plugin itself, upsamplers/basic_upsamplers.py:28-33)
//              followed by ConvModule(3x3, pad 1, bias) + ReLU                        core/model/heads/conv_heads.py:59-73
//
// The resized map Y[p] = sum_{q in 2x2(p)} a(p, q) X[q] is a blend of a (H/h)^2 times smaller map, and the convolution is
// linear in it, so
//     out[p][n] = act(b[n] + sum_t [p + t inside] sum_c W_t[n][c] Y[p + t][c])
//               = act(b[n] + sum_t [p + t inside] sum_{q in 2x2(p + t)} a(p + t, q) Z_t[q][n]),     Z_t = X W_t^T  (low resolution).
// Z = X [B h w, C] x [W_0 .. W_8]^T is ONE dense GEMM at low resolution ([B h w, 9 N], column t N + n; isp_gemm_f16) and
// the kernel below is the blend: 36 multiply-adds per output value instead of 9 C (= 3 456 at C = 384, 9 216 at C = 1024),
// and the [B, H, W, C] map (4.9 GB at batch 32 x 448^2 x 384, 1.6 GB per image at 896^2 x 1024) never exists.
//
// Kernels: a workgroup owns a 16 x 16 patch of output pixels.  The source pixels its 18 x 18 tap neighbourhood touches
// (<= FMAX x FMAX, checked by the launcher with the kernel's own fp32 coordinate arithmetic) are staged per 64-channel block in LDS
// as [q][tap][channels]; tap validity is folded into the blend weights as zeros with clamped coordinates, so border tiles run
// the same code.  Two forms below: on the matrix pipe for half tap planes (the product path), on the vector pipe for fp32 ones
// (the checking mode).  HBM sees the output map once (4.93 GB at batch 32 x 448^2 x 384: 1.70 ms; a plain resize kernel writing
// the same map takes 1.71 ms, a broadcast multiply 1.30 ms), Z stays in L2 / Infinity Cache.
#include <type_traits>

#include "isp_common.h"

namespace {

constexpr int FMAX = 5;      // staged source footprint per axis
constexpr int TPX = 16;      // output tile edge

template <typename ZT>
struct ZTraits;
template <>
struct ZTraits<_Float16> {
    static constexpr int CB = 64;  // channels per block
};
template <>
struct ZTraits<float> {
    static constexpr int CB = 32;
};

template <int OUT>
__device__ __forceinline__ void store8(void* out, size_t idx, const float* v) {
    if constexpr (OUT == ISP_F32) {
        float4* o = reinterpret_cast<float4*>(reinterpret_cast<float*>(out) + idx);
        o[0] = make_float4(v[0], v[1], v[2], v[3]);
        o[1] = make_float4(v[4], v[5], v[6], v[7]);
    } else if constexpr (OUT == ISP_F16) {
        *reinterpret_cast<uint4*>(reinterpret_cast<unsigned short*>(out) + idx) =
            make_uint4(pack2h_sat(v[0], v[1]), pack2h_sat(v[2], v[3]), pack2h_sat(v[4], v[5]), pack2h_sat(v[6], v[7]));
    } else {
        *reinterpret_cast<uint4*>(reinterpret_cast<unsigned short*>(out) + idx) =
            make_uint4(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]), pack2bf(v[4], v[5]), pack2bf(v[6], v[7]));
    }
}

// src coordinate of an output coordinate, exactly as the resize kernels (and torch) compute it: fp32 product, truncation
__device__ __host__ __forceinline__ void src_coord(int d, float s, int n_in, int& i0, int& i1, float& l) {
    const float f = s * (float)d;
    i0 = (int)f;
    i1 = i0 + 1 < n_in ? i0 + 1 : n_in - 1;
    l = f - (float)i0;
}

// ---- vector form (fp32 tap planes: the checking mode; ISEGPROBE_BLEND_FORM=3 for half).  The blend is separable,
// a(p + t, q) = ay(Y + ty, qy) ax(X + tx, qx), so a tile first combines the tap planes ALONG X at the source rows it touches,
//     U[ty][qy][X][n] = sum_tx [X + tx inside] sum_{qx in 2(X + tx)} ax(X + tx, qx) Z[qy][qx][(ty, tx)][n]      (6 multiply-adds),
// into LDS (3 x fy x 16 entries per channel) and then blends ALONG Y per output pixel (6 more): 12 multiply-adds and LDS reads
// per value instead of 36 (a tile's U entries serve its 16 rows).  A lane owns ONE 16-byte channel group (8 half / 4 fp32
// channels) of an entry or pixel (the first forms gave a thread all 64 channels of a pixel: 3.13 ms one-phase, 2.17 ms two-phase,
// against 1.90 ms for this one and 1.70 ms for the matrix-pipe form below, batch 32 x 448^2 x 384).  A thread = (group g of 32, octet o of 8); phase A computes column g % 16 of the U rows
// g / 16, g / 16 + 2, .., phase B row g % 16 of the pixel columns g / 16, g / 16 + 2, ..: every address is a lane constant plus a
// loop constant, and eight neighbouring lanes read / write one aligned 128-byte row, so
//   * the output leaves as whole 128-byte lines (a thread-per-pixel store instruction touches 64 different lines: its address
//     processing alone was a quarter of the kernel), U and Z reads are conflict-free without padding U's rows,
//   * phase A keeps all 256 threads busy (144 entries at a 3 x 3 footprint used 2.25 of 4 waves),
//   * a thread holds 8 accumulators instead of 64 (the opaque-asm fences of the first forms are not needed), and
//   * the per-column / per-row blend weights and source indices are tile-wide tables in LDS (16 x 6 each), read as broadcasts.
template <typename ZT, int OUT, bool RELU>
__global__ __launch_bounds__(256) void conv_bilinear_blend3_kernel(const ZT* __restrict__ z, const float* __restrict__ bias,
                                                                   void* __restrict__ out, int h, int w, int H, int W, int N,
                                                                   float sy, float sx, int zs_bytes) {
    constexpr int CB = ZTraits<ZT>::CB;
    constexpr int ES = (int)sizeof(ZT);
    constexpr int VEC = 16 / ES;                    // channels per lane
    static_assert(CB / VEC == 8, "eight 16-byte groups per entry");
    constexpr int QROW = 9 * CB * ES, QPITCH = QROW + 16, PIECES = QROW / 16, TAPB = CB * ES;
    constexpr int UROW = CB * ES;                   // 128 bytes per (ty, qy, X) entry, unpadded: two neighbouring entries = 64 banks
    extern __shared__ __attribute__((aligned(16))) char lds[];
    char* const zs = lds;
    char* const us = lds + zs_bytes;
    __shared__ float xw[TPX][6], yw[TPX][6];        // blend weights per tile column / row: [tap][corner], 0 where the tap leaves the image
    __shared__ int xi[TPX][6], yi[TPX][6];          // ... and the corner's index inside the staged footprint

    const int tid = threadIdx.x;
    const int o = tid & 7, g = tid >> 3;
    const int b = blockIdx.z;
    const int Y0 = blockIdx.y * TPX, X0 = blockIdx.x * TPX;
    int qy_lo, qx_lo, fy, fx;
    {
        int i0, i1;
        float l;
        src_coord(max(Y0 - 1, 0), sy, h, i0, i1, l);
        qy_lo = i0;
        src_coord(min(Y0 + TPX, H - 1), sy, h, i0, i1, l);
        fy = i1 - qy_lo + 1;
        src_coord(max(X0 - 1, 0), sx, w, i0, i1, l);
        qx_lo = i0;
        src_coord(min(X0 + TPX, W - 1), sx, w, i0, i1, l);
        fx = i1 - qx_lo + 1;
    }
    if (tid < 2 * TPX) {  // threads 0-15: column table, 16-31: row table
        const bool col = tid < TPX;
        const int c = col ? tid : tid - TPX;
        const int base = (col ? X0 : Y0) + c, n_out = col ? W : H, n_in = col ? w : h, lo = col ? qx_lo : qy_lo, f = col ? fx : fy;
        const float sc = col ? sx : sy;
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            int i0, i1;
            float l;
            const int d = base + t - 1;
            const bool v = d >= 0 && d < n_out;
            src_coord(min(max(d, 0), n_out - 1), sc, n_in, i0, i1, l);
            i0 = min(max(i0 - lo, 0), f - 1), i1 = min(max(i1 - lo, 0), f - 1);
            (col ? xw : yw)[c][2 * t] = v ? 1.f - l : 0.f, (col ? xw : yw)[c][2 * t + 1] = v ? l : 0.f;
            (col ? xi : yi)[c][2 * t] = i0, (col ? xi : yi)[c][2 * t + 1] = i1;
        }
    }
    const size_t zrow = (size_t)9 * N;
    const ZT* zb = z + ((size_t)b * h * w) * zrow;
    const int nq = fy * fx, n_rows = 3 * fy;  // U rows R = ty * fy + qy, 16 entries (X) each
    __syncthreads();                          // tables
    // Lane-constant address parts, so that the inner loops are one ds_read_b128 + 8 multiply-adds per tap:
    //   phase A: this lane's column X = g % 16 for every entry it handles (rows R = g / 16 + 2 j): Z address = rowA[j] + colA[k];
    //   phase B: this lane's row py = g % 16 for every pixel it handles (columns g / 16 + 2 k): U address = rowB[j] + immediate.
    const int ax = g & 15, ar0 = g >> 4;
    int colA[6];
    float wxa[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) colA[k] = xi[ax][k] * QPITCH + (k >> 1) * TAPB + o * 16, wxa[k] = xw[ax][k];
    const int py = g & 15, bx0 = g >> 4;
    int rowB[6];
    float wyb[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) rowB[j] = (((j >> 1) * fy + yi[py][j]) * TPX + bx0) * UROW + o * 16, wyb[j] = yw[py][j];
    const int Y = Y0 + py;
    const size_t orow = (((size_t)b * H + Y) * W + X0 + bx0) * N + o * VEC;  // pixel (py, bx0); + 2 k N per step

    auto fma_group = [&](float (&acc)[VEC], const char* p, float wv) {
        if constexpr (ES == 2) {
            const f16x8_t v = *reinterpret_cast<const f16x8_t*>(p);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] = fmaf((float)v[e], wv, acc[e]);
        } else {
            const float4 v = *reinterpret_cast<const float4*>(p);
            acc[0] = fmaf(v.x, wv, acc[0]), acc[1] = fmaf(v.y, wv, acc[1]), acc[2] = fmaf(v.z, wv, acc[2]), acc[3] = fmaf(v.w, wv, acc[3]);
        }
    };

    for (int n0 = 0; n0 < N; n0 += CB) {
        // (no barrier here: zs is free since the barrier behind the previous phase A, and us is written only behind the next one,
        //  which every thread reaches after its previous phase B)
        for (int i = tid; i < nq * PIECES; i += 256) {
            const int q = i / PIECES, pc = i - q * PIECES;
            const int t = pc / (PIECES / 9), r = pc - t * (PIECES / 9);
            const int qy = qy_lo + q / fx, qx = qx_lo + q % fx;
            const ZT* src = zb + ((size_t)qy * w + qx) * zrow + (size_t)t * N + n0 + r * VEC;
            *reinterpret_cast<uint4*>(zs + q * QPITCH + pc * 16) = *reinterpret_cast<const uint4*>(src);
        }
        __syncthreads();
        // ---- phase A: U[R][X = ax] (this lane's channel group) for the rows R = ar0, ar0 + 2, ..
        for (int R = ar0; R < n_rows; R += 2) {
            const int ty = R / fy, qy = R - ty * fy;
            const char* zr = zs + qy * fx * QPITCH + ty * 3 * TAPB;
            float acc[VEC];
#pragma unroll
            for (int e = 0; e < VEC; ++e) acc[e] = 0.f;
#pragma unroll
            for (int k = 0; k < 6; ++k) fma_group(acc, zr + colA[k], wxa[k]);
            char* up = us + (R * TPX + ax) * UROW + o * 16;
            if constexpr (ES == 2)
                *reinterpret_cast<uint4*>(up) = make_uint4(pack2h_sat(acc[0], acc[1]), pack2h_sat(acc[2], acc[3]), pack2h_sat(acc[4], acc[5]),
                                                           pack2h_sat(acc[6], acc[7]));
            else
                *reinterpret_cast<float4*>(up) = make_float4(acc[0], acc[1], acc[2], acc[3]);
        }
        __syncthreads();
        // ---- phase B: pixels (py, bx0 + 2 k): six U entries each, at immediate offsets from this lane's six row addresses
        float bv[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) bv[e] = bias ? bias[n0 + o * VEC + e] : 0.f;
#pragma unroll
        for (int k = 0; k < TPX / 2; ++k) {
            float acc[VEC];
#pragma unroll
            for (int e = 0; e < VEC; ++e) acc[e] = 0.f;
#pragma unroll
            for (int j = 0; j < 6; ++j) fma_group(acc, us + rowB[j] + 2 * k * UROW, wyb[j]);
            if (Y < H && X0 + bx0 + 2 * k < W) {
                const size_t at = orow + (size_t)2 * k * N + n0;
#pragma unroll
                for (int e = 0; e < VEC; ++e) {
                    acc[e] += bv[e];
                    if (RELU) acc[e] = fmaxf(acc[e], 0.f);
                }
                if constexpr (OUT == ISP_F32) {
                    float* op = reinterpret_cast<float*>(out) + at;
#pragma unroll
                    for (int e = 0; e < VEC; e += 4) *reinterpret_cast<float4*>(op + e) = make_float4(acc[e], acc[e + 1], acc[e + 2], acc[e + 3]);
                } else if constexpr (VEC == 8) {
                    unsigned short* op = reinterpret_cast<unsigned short*>(out) + at;
                    if constexpr (OUT == ISP_F16)
                        *reinterpret_cast<uint4*>(op) = make_uint4(pack2h_sat(acc[0], acc[1]), pack2h_sat(acc[2], acc[3]),
                                                                   pack2h_sat(acc[4], acc[5]), pack2h_sat(acc[6], acc[7]));
                    else
                        *reinterpret_cast<uint4*>(op) = make_uint4(pack2bf(acc[0], acc[1]), pack2bf(acc[2], acc[3]), pack2bf(acc[4], acc[5]),
                                                                   pack2bf(acc[6], acc[7]));
                }
            }
        }
    }
}

// ---- the blend on the matrix pipe (half tap planes; the default for them).  PMC of the vector form above: 7 163 vector
// instructions per wave, vector pipe 70 % busy, 2.6 of the 6.9 TB/s a plain fill writes here -- the 12 multiply-adds per value ARE
// the kernel.  Both phases are small matrix products against tile-constant weight matrices with K <= 15:
//   phase A, per U row R = (ty, qy):   U^T[ch][X]   = sum_{k = (tx, qx)} Z^T[ch][k] Bx[k][X]      K = 3 tx x (fx <= 5 source columns)
//   phase B, per pixel column px:       out^T[ch][py] = sum_{k = R}       U^T[ch][k] Ay[k][py]      K = 3 ty x (fy <= 5 source rows)
// on v_mfma_f32_16x16x16_f16 (M = 16 channels, N = 16 columns / rows of the tile).  The A operand's rows k are rows of the
// staged images at arbitrary addresses -- exactly what ds_read_b64_tr_b16 gathers (each lane of a 16-lane group supplies the
// address of one row's four channels; lane i receives channel i of the group's four rows) -- and B is a per-lane constant
// built once per tile (the bilinear weights, rounded to half: 2^-12 like the tap planes themselves).  The four MFMAs of a
// 64-channel block take channels 32 (c / 2) + 8 p + 4 (c % 2) + {0..3} from address lane p, so a lane ends with two runs of 8
// consecutive channels of its column / pixel and the four lanes of a pixel store 64 consecutive bytes per instruction.  Per wave and channel block:
// ~25 MFMAs + 25 transposed reads + the epilogue of 4 x 16 values, against 1 194 vector instructions.
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
__device__ __forceinline__ f16x4_t tr_read4(const char* p) {
    return __builtin_bit_cast(f16x4_t, __builtin_amdgcn_ds_read_tr16_b64_v4i16((ISP_LDS s16x4_t*)p));
}

template <int OUT, bool RELU>
__global__ __launch_bounds__(256) void conv_bilinear_blend4_kernel(const _Float16* __restrict__ z, const float* __restrict__ bias,
                                                                   void* __restrict__ out, int h, int w, int H, int W, int N,
                                                                   float sy, float sx, int zs_bytes) {
    constexpr int CB = 64, ES = 2, VEC = 8;
    constexpr int UROW = CB * ES;
    // one 64-channel block per staging pass, [q][tap][64 channels] (two blocks per pass -- half the passes, 46 KiB of LDS --
    // measured 1.97 against 1.76 ms)
    constexpr int TAPB = CB * ES, QPITCH = 9 * TAPB + 16, PIECES = 9 * TAPB / 16;
#ifndef ISP_BLEND_ABLATE  // timing builds only (-DISP_BLEND_ABLATE=1: no stores, 2: no staging, 3: neither); such a library computes garbage
#define ISP_BLEND_ABLATE 0
#endif
    constexpr int abl = ISP_BLEND_ABLATE;
    constexpr int KF = 5;  // k = tap * KF + source index inside the footprint (<= 5 per axis): 15 rows, row 15 is a zero-weight dummy
    extern __shared__ __attribute__((aligned(16))) char lds[];
    char* const zs = lds;
    char* const us = lds + zs_bytes;
    __shared__ float xw[TPX][6], yw[TPX][6];
    __shared__ int xi[TPX][6], yi[TPX][6];

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int b = blockIdx.z;
    const int Y0 = blockIdx.y * TPX, X0 = blockIdx.x * TPX;
    int qy_lo, qx_lo, fy, fx;
    {
        int i0, i1;
        float l;
        src_coord(max(Y0 - 1, 0), sy, h, i0, i1, l);
        qy_lo = i0;
        src_coord(min(Y0 + TPX, H - 1), sy, h, i0, i1, l);
        fy = i1 - qy_lo + 1;
        src_coord(max(X0 - 1, 0), sx, w, i0, i1, l);
        qx_lo = i0;
        src_coord(min(X0 + TPX, W - 1), sx, w, i0, i1, l);
        fx = i1 - qx_lo + 1;
    }
    if (tid < 2 * TPX) {  // threads 0-15: column table, 16-31: row table (as in the vector form)
        const bool col = tid < TPX;
        const int c = col ? tid : tid - TPX;
        const int base = (col ? X0 : Y0) + c, n_out = col ? W : H, n_in = col ? w : h, lo = col ? qx_lo : qy_lo, f = col ? fx : fy;
        const float sc = col ? sx : sy;
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            int i0, i1;
            float l;
            const int d = base + t - 1;
            const bool v = d >= 0 && d < n_out;
            src_coord(min(max(d, 0), n_out - 1), sc, n_in, i0, i1, l);
            i0 = min(max(i0 - lo, 0), f - 1), i1 = min(max(i1 - lo, 0), f - 1);
            (col ? xw : yw)[c][2 * t] = v ? 1.f - l : 0.f, (col ? xw : yw)[c][2 * t + 1] = v ? l : 0.f;
            (col ? xi : yi)[c][2 * t] = i0, (col ? xi : yi)[c][2 * t + 1] = i1;
        }
    }
    __syncthreads();
    // ---- tile constants of this lane.  MFMA lane roles: n = lane % 16 (tile column X in phase A, tile row py in phase B),
    // k = 4 (lane / 16) + e for B-operand element e; as an address lane of a transposed read: group gq = lane / 16 reads rows
    // k = 4 gq + q with q = (lane % 16) / 4, channels 16 p + 4 c + {0..3} with p = lane % 4.
    const int n = lane & 15, gq = lane >> 4, aq = (lane & 15) >> 2, ap = lane & 3;
    f16x4_t bxa, bya;  // B operands: Bx[k][X = n], Ay[k][py = n]
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int k = 4 * gq + e, t = k / KF, s_ = k - t * KF;
        float vx = 0.f, vy = 0.f;
        if (t < 3) {
            vx = (xi[n][2 * t] == s_ ? xw[n][2 * t] : 0.f) + (xi[n][2 * t + 1] == s_ ? xw[n][2 * t + 1] : 0.f);
            vy = (yi[n][2 * t] == s_ ? yw[n][2 * t] : 0.f) + (yi[n][2 * t + 1] == s_ ? yw[n][2 * t + 1] : 0.f);
        }
        bxa[e] = (_Float16)vx, bya[e] = (_Float16)vy;
    }
    // the row this lane ADDRESSES in the transposed reads: k = 4 gq + aq
    const int ak = 4 * gq + aq, at = ak / KF, as_ = ak - at * KF;
    const bool a_on = at < 3;
    // phase A: Z row (tx = at, source column as_) of U row R = (ty, qy): zs + (qy fx + as_) QPITCH + (ty 3 + at) TAPB
    const int za = (a_on ? min(as_, fx - 1) * QPITCH + at * TAPB : 0) + ap * 16;
    // phase B: U row (ty = at, source row as_) at pixel column px: us + ((at fy + as_) 16 + px) UROW
    const int ua = (a_on ? (at * fy + min(as_, fy - 1)) * TPX * UROW : 0) + ap * 16;
    const size_t zrow = (size_t)9 * N;
    const _Float16* zb = z + ((size_t)b * h * w) * zrow;
    const int nq = fy * fx, n_rows = 3 * fy;
    const int Y = Y0 + n;  // phase B: this lane's pixel row

    for (int n0 = 0; n0 < N; n0 += CB) {
        // no barrier here: every wave is past the barrier behind the previous block's phase A, so zs is free, and a wave that
        // finishes its phase B early starts fetching the next block while the others still store
        if (!(abl & 2))
            for (int i = tid; i < nq * PIECES; i += 256) {
                const int q = i / PIECES, pc = i - q * PIECES;
                const int t = pc / (PIECES / 9), r = pc - t * (PIECES / 9);
                const int qy = qy_lo + q / fx, qx = qx_lo + q % fx;
                const _Float16* src = zb + ((size_t)qy * w + qx) * zrow + (size_t)t * N + n0 + r * VEC;
                *reinterpret_cast<uint4*>(zs + q * QPITCH + pc * 16) = *reinterpret_cast<const uint4*>(src);
            }
        __syncthreads();  // stage landed; the previous block's phase B is done with us
        // ---- phase A: wave `wid` forms the U rows R = wid, wid + 4, ..
        for (int R = wid; R < n_rows; R += 4) {
            const int ty = R / fy, qy = R - ty * fy;
            const char* zr = zs + qy * fx * QPITCH + ty * 3 * TAPB + za;
            f32x4 d[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const f16x4_t a = tr_read4(zr + (c >> 1) * 64 + (c & 1) * 8);  // channels 32 (c / 2) + 8 p + 4 (c % 2) .. + 3 of the lane's row
                d[c] = __builtin_amdgcn_mfma_f32_16x16x16f16(a, bxa, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
            }
            // D^T[m = 4 gq + i][X = n] of MFMA c = channel 32 (c / 2) + 8 gq + 4 (c % 2) + i: two runs of 8 channels of U[R][X = n]
            char* up = us + (R * TPX + n) * UROW + gq * 16;
            *reinterpret_cast<uint4*>(up) = make_uint4(pack2h_sat(d[0][0], d[0][1]), pack2h_sat(d[0][2], d[0][3]),
                                                       pack2h_sat(d[1][0], d[1][1]), pack2h_sat(d[1][2], d[1][3]));
            *reinterpret_cast<uint4*>(up + 64) = make_uint4(pack2h_sat(d[2][0], d[2][1]), pack2h_sat(d[2][2], d[2][3]),
                                                            pack2h_sat(d[3][0], d[3][1]), pack2h_sat(d[3][2], d[3][3]));
        }
        __syncthreads();
        // ---- phase B: wave `wid` forms the pixel columns px = wid, wid + 4, wid + 8, wid + 12
        float bv[16];  // channels n0 + 8 gq + {0..7} and n0 + 32 + 8 gq + {0..7}
#pragma unroll
        for (int e = 0; e < 16; ++e) bv[e] = bias ? bias[n0 + (e >> 3) * 32 + gq * 8 + (e & 7)] : 0.f;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int px = wid + 4 * kk;
            const char* ur = us + px * UROW + ua;
            f32x4 d[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const f16x4_t a = tr_read4(ur + (c >> 1) * 64 + (c & 1) * 8);
                d[c] = __builtin_amdgcn_mfma_f32_16x16x16f16(a, bya, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
            }
            const int X = X0 + px;
            // (abl & 1: no stores.  The second clause is always true but opaque to the compiler, so the arithmetic above stays.)
            if (Y < H && X < W && !((abl & 1) && n0 + (int)blockIdx.x < 1000000)) {
                float v[16];
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        float r = d[c][i] + bv[4 * c + i];
                        v[4 * c + i] = RELU ? fmaxf(r, 0.f) : r;
                    }
                // the four lanes of a pixel write 64 consecutive bytes per instruction (whole sectors), the two instructions its line
                const size_t at_ = (((size_t)b * H + Y) * W + X) * N + n0 + gq * 8;
                store8<OUT>(out, at_, v);
                store8<OUT>(out, at_ + 32, v + 8);
            }
        }
    }
}

// largest per-axis footprint over the tiles, with the kernel's own arithmetic
int max_footprint(int n_in, int n_out, float s) {
    int worst = 0;
    for (int t0 = 0; t0 < n_out; t0 += TPX) {
        int lo, hi, i1;
        float l;
        src_coord(t0 - 1 > 0 ? t0 - 1 : 0, s, n_in, lo, i1, l);
        src_coord(t0 + TPX < n_out - 1 ? t0 + TPX : n_out - 1, s, n_in, hi, i1, l);
        if (i1 - lo + 1 > worst) worst = i1 - lo + 1;
    }
    return worst;
}

template <typename ZT, int OUT>
int launch(const void* z, const float* bias, void* out, int B, int h, int w, int H, int W, int N, int relu, hipStream_t s) {
    const float sy = H > 1 ? (float)(h - 1) / (float)(H - 1) : 0.f;
    const float sx = W > 1 ? (float)(w - 1) / (float)(W - 1) : 0.f;
    const dim3 grid((W + TPX - 1) / TPX, (H + TPX - 1) / TPX, B);
    // ISEGPROBE_BLEND_FORM: 4 (default for half tap planes) both phases on MFMA; 3 (fp32 tap planes; A/B switch for half) the vector form
    static const int form_env = [] { const char* e = getenv("ISEGPROBE_BLEND_FORM"); return e ? atoi(e) : 4; }();
    const int form = (form_env == 4 && !std::is_same<ZT, _Float16>::value) ? 3 : form_env;  // 4 = matrix-pipe form, half tap planes only
    constexpr int CB = ZTraits<ZT>::CB, ES = (int)sizeof(ZT);
    const int FY = max_footprint(h, H, sy), FX = max_footprint(w, W, sx);
    const int zs_bytes = FY * FX * (9 * CB * ES + 16);
    if (form == 4) {
        if constexpr (std::is_same<ZT, _Float16>::value) {
            const int lds = zs_bytes + 3 * FY * TPX * (CB * ES);
            if (relu)
                conv_bilinear_blend4_kernel<OUT, true><<<grid, 256, lds, s>>>((const _Float16*)z, bias, out, h, w, H, W, N, sy, sx, zs_bytes);
            else
                conv_bilinear_blend4_kernel<OUT, false><<<grid, 256, lds, s>>>((const _Float16*)z, bias, out, h, w, H, W, N, sy, sx, zs_bytes);
        }
    } else {
        const int lds = zs_bytes + 3 * FY * TPX * (CB * ES);  // 3 x 3 footprint: 10.5 + 18 KiB; 5 x 5: 29.2 + 30 KiB
        if (relu)
            conv_bilinear_blend3_kernel<ZT, OUT, true><<<grid, 256, lds, s>>>((const ZT*)z, bias, out, h, w, H, W, N, sy, sx, zs_bytes);
        else
            conv_bilinear_blend3_kernel<ZT, OUT, false><<<grid, 256, lds, s>>>((const ZT*)z, bias, out, h, w, H, W, N, sy, sx, zs_bytes);
    }
    return isp_launch_status();
}

// ---- adjoint of the blend (training): dZ_t[q][n] = sum_p [p + t inside] a(p + t, q) g[p][n] for the (ReLU-masked) output
// gradient g [B, H, W, N].  Separable like the forward: along Y into V[b][qy][ty][X][n] (fp32 workspace), then along X into
// dZ[b h w][(ty, tx) N + n].  Both kernels are gathers -- a thread owns one output entry (8 channels) and walks the ~2 / s
// pixels whose bilinear footprint contains its source index, with the weight the forward used, (i0 == q ? 1 - l : 0) +
// (i1 == q ? l : 0) from the same fp32 coordinate arithmetic; one load serves the three taps of the walked axis -- so nothing
// is atomic and the result does not depend on the launch.  Memory-bound: g is read three times from L2 (once per row tap).
__device__ __forceinline__ float tent(int d, float s, int n_in, int n_out, int q) {  // weight of source index q at output coordinate d
    if (d < 0 || d >= n_out) return 0.f;
    int i0, i1;
    float l;
    src_coord(d, s, n_in, i0, i1, l);
    return (i0 == q ? 1.f - l : 0.f) + (i1 == q ? l : 0.f);
}
// output coordinates whose footprint can contain source index q: s d in (q - 1, q + 1)
__device__ __forceinline__ void support(int q, float s, int n_out, int& lo, int& hi) {
    if (s <= 0.f) {
        lo = 0, hi = n_out - 1;
        return;
    }
    lo = (int)floorf((float)(q - 1) / s) - 1, hi = (int)ceilf((float)(q + 1) / s) + 1;  // (one pixel of slack for the rounding)
    lo = lo < 0 ? 0 : lo, hi = hi > n_out - 1 ? n_out - 1 : hi;
}

__global__ __launch_bounds__(256) void conv_bilinear_adj_y_kernel(const bf16_t* __restrict__ g, float* __restrict__ V, int h,
                                                                  int H, int W, int N, float sy) {
    const int n8 = N >> 3;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)W * n8) return;
    const int qy = blockIdx.y, b = blockIdx.z;
    int lo, hi;
    support(qy, sy, H, lo, hi);
    float acc[3][8];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[t][e] = 0.f;
    const bf16_t* gp = g + ((size_t)b * H * W) * N + idx * 8;
    // output row y feeds tap ty through the resized-map row y + ty - 1
    for (int y = (lo - 1 < 0 ? 0 : lo - 1); y <= (hi + 1 > H - 1 ? H - 1 : hi + 1); ++y) {
        const float a0 = tent(y - 1, sy, h, H, qy), a1 = tent(y, sy, h, H, qy), a2 = tent(y + 1, sy, h, H, qy);
        if (a0 == 0.f && a1 == 0.f && a2 == 0.f) continue;
        const uint4 v = *reinterpret_cast<const uint4*>(gp + (size_t)y * W * N);
        const unsigned* q = &v.x;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float x0 = __uint_as_float(q[e] << 16), x1 = __uint_as_float(q[e] & 0xffff0000u);
            acc[0][2 * e] += a0 * x0, acc[0][2 * e + 1] += a0 * x1;
            acc[1][2 * e] += a1 * x0, acc[1][2 * e + 1] += a1 * x1;
            acc[2][2 * e] += a2 * x0, acc[2][2 * e + 1] += a2 * x1;
        }
    }
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        float* vp = V + ((((size_t)b * h + qy) * 3 + t) * W) * N + idx * 8;
        *reinterpret_cast<float4*>(vp) = make_float4(acc[t][0], acc[t][1], acc[t][2], acc[t][3]);
        *reinterpret_cast<float4*>(vp + 4) = make_float4(acc[t][4], acc[t][5], acc[t][6], acc[t][7]);
    }
}

__global__ __launch_bounds__(256) void conv_bilinear_adj_x_kernel(const float* __restrict__ V, bf16_t* __restrict__ dz, int h,
                                                                  int w, int W, int N, float sx, long total) {
    const int n8 = N >> 3;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;  // ((b h + qy) w + qx) * 3 + ty) * n8 + c8
    if (idx >= total) return;
    const int c8 = (int)(idx % n8);
    long r = idx / n8;
    const int ty = (int)(r % 3);
    r /= 3;
    const int qx = (int)(r % w);
    const long bq = r / w;  // b h + qy
    int lo, hi;
    support(qx, sx, W, lo, hi);
    float acc[3][8];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[t][e] = 0.f;
    const float* vp = V + (((size_t)bq * 3 + ty) * W) * N + c8 * 8;
    for (int x = (lo - 1 < 0 ? 0 : lo - 1); x <= (hi + 1 > W - 1 ? W - 1 : hi + 1); ++x) {
        const float a0 = tent(x - 1, sx, w, W, qx), a1 = tent(x, sx, w, W, qx), a2 = tent(x + 1, sx, w, W, qx);
        if (a0 == 0.f && a1 == 0.f && a2 == 0.f) continue;
        const float4 u0 = *reinterpret_cast<const float4*>(vp + (size_t)x * N), u1 = *reinterpret_cast<const float4*>(vp + (size_t)x * N + 4);
        const float u[8] = {u0.x, u0.y, u0.z, u0.w, u1.x, u1.y, u1.z, u1.w};
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[0][e] += a0 * u[e], acc[1][e] += a1 * u[e], acc[2][e] += a2 * u[e];
    }
    bf16_t* zp = dz + ((size_t)(r)) * 9 * N + (size_t)(ty * 3) * N + c8 * 8;  // r = (b h + qy) w + qx
#pragma unroll
    for (int t = 0; t < 3; ++t)
        *reinterpret_cast<uint4*>(zp + (size_t)t * N) = make_uint4(pack2bf(acc[t][0], acc[t][1]), pack2bf(acc[t][2], acc[t][3]),
                                                                   pack2bf(acc[t][4], acc[t][5]), pack2bf(acc[t][6], acc[t][7]));
}

}  // namespace

extern "C" int isp_conv3x3_of_bilinear_supported(int h, int w, int H, int W, int N, int z_dtype) {
    if (h <= 0 || w <= 0 || H <= 0 || W <= 0 || N <= 0) return 0;
    if (z_dtype != ISP_F16 && z_dtype != ISP_F32) return 0;
    if (N % (z_dtype == ISP_F16 ? 64 : 32)) return 0;
    const float sy = H > 1 ? (float)(h - 1) / (float)(H - 1) : 0.f;
    const float sx = W > 1 ? (float)(w - 1) / (float)(W - 1) : 0.f;
    return max_footprint(h, H, sy) <= FMAX && max_footprint(w, W, sx) <= FMAX;
}

extern "C" int isp_conv3x3_of_bilinear_blend(const void* z, int z_dtype, const float* bias, void* out, int out_dtype, int B,
                                             int h, int w, int H, int W, int N, int relu, void* stream) {
    ISP_CHECK_ARG(z && out && B > 0 && B <= 65535 && h > 0 && w > 0 && H > 0 && W > 0 && N > 0);
    ISP_CHECK_ARG(((uintptr_t)z & 15) == 0 && ((uintptr_t)out & 15) == 0);
    if (!isp_conv3x3_of_bilinear_supported(h, w, H, W, N, z_dtype)) return ISP_ERR_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    if (z_dtype == ISP_F16) {
        if (out_dtype == ISP_F16) return launch<_Float16, ISP_F16>(z, bias, out, B, h, w, H, W, N, relu, s);
        if (out_dtype == ISP_BF16) return launch<_Float16, ISP_BF16>(z, bias, out, B, h, w, H, W, N, relu, s);
        if (out_dtype == ISP_F32) return launch<_Float16, ISP_F32>(z, bias, out, B, h, w, H, W, N, relu, s);
    } else {
        if (out_dtype == ISP_F32) return launch<float, ISP_F32>(z, bias, out, B, h, w, H, W, N, relu, s);
    }
    return ISP_ERR_UNSUPPORTED;
}

extern "C" long isp_conv3x3_of_bilinear_bwd_workspace_bytes(int B, int h, int W, int N) {
    if (B <= 0 || h <= 0 || W <= 0 || N <= 0) return ISP_ERR_INVALID;
    return (long)B * h * 3 * W * N * 4;
}

extern "C" int isp_conv3x3_of_bilinear_blend_bwd(const void* g_bf16, void* dz_bf16, void* workspace, int B, int h, int w, int H,
                                                 int W, int N, void* stream) {
    ISP_CHECK_ARG(g_bf16 && dz_bf16 && workspace && B > 0 && B <= 65535 && h > 0 && h <= 65535 && w > 0 && H > 0 && W > 0 && N > 0 && N % 8 == 0);
    ISP_CHECK_ARG(((uintptr_t)g_bf16 & 15) == 0 && ((uintptr_t)dz_bf16 & 15) == 0 && ((uintptr_t)workspace & 15) == 0);
    const float sy = H > 1 ? (float)(h - 1) / (float)(H - 1) : 0.f;
    const float sx = W > 1 ? (float)(w - 1) / (float)(W - 1) : 0.f;
    hipStream_t s = (hipStream_t)stream;
    const long cols = (long)W * (N / 8);
    conv_bilinear_adj_y_kernel<<<dim3((unsigned)((cols + 255) / 256), h, B), 256, 0, s>>>((const bf16_t*)g_bf16, (float*)workspace, h,
                                                                                         H, W, N, sy);
    const long total = (long)B * h * w * 3 * (N / 8);
    conv_bilinear_adj_x_kernel<<<(unsigned)((total + 255) / 256), 256, 0, s>>>((const float*)workspace, (bf16_t*)dz_bf16, h, w, W, N, sx,
                                                                              total);
    return isp_launch_status();
}
