// Weight gradient of a 3x3 / stride 1 / pad 1 convolution, all nine taps in one pass:
//     dW[n][ky][kx][c] += sum_{b,y,x} g[b,y,x,n] * x[b, y+ky-1, x+kx-1, c]        (zero outside the image)
// g = gradient of the conv's pre-activation [B,H,W,N] bf16, x = the conv's input [B,H,W,C] bf16, dW fp32
// [N][9*C] (the layout of conv.weight.permute(0,2,3,1)), accumulated with atomics into a caller-zeroed buffer.
//
// What autograd computes for ConvModule's weight in the reference's training step (heads/conv_heads.py:51-73
// under trainer.py:219-226).  The first build ran nine pixel-reduction GEMMs (isp_tn_gemm_bf16_atomic, one per
// tap): each re-streamed g and x from L2, 66 GB of L2->LDS traffic per conv at batch 8 / 448^2, and sat on the
// L2 bandwidth at 465 TFLOP/s.  Here a block stages an 8x8-pixel tile of g and the matching 10x10 input patch
// once and forms all nine taps from it (3.5x less staging per FLOP).
//
//   block  = 8 waves, output tile 64 input channels x 128 output channels x 9 taps, looping over the 8x8-pixel
//            tiles of its share of the images (split-K over pixels; fp32 atomics at the end);
//   wave   = 16 c x 64 n x 9 taps = 36 accumulator tiles (144 VGPRs) of v_mfma_f32_16x16x32_bf16, contraction
//            over 32 pixels (4 tile rows x 8 pixels);
//   both operands are pixel-major in HBM and in LDS, so every fragment (8 pixels of one channel) is two
//   ds_read_b64_tr_b16; the g fragments of a k-step are shared by the nine taps, the x fragments are the same
//   patch read at nine shifted positions;
//   LDS images are filled by LDS-DMA: g rows are 256 B (128 n) with an XOR swizzle on the source address; patch
//   rows are 128 B of data at a pitch of 160 B (12 pixels per patch row): the 8 pixel rows a half-wave's
//   transposed read touches (p..p+3 and p+12..p+15) then start 40 dwords apart mod 64, i.e. in 8 distinct 32-byte
//   windows for every tap, with NO swizzle -- so each of the 36 shifted fragment addresses per tile is the lane's
//   base plus a compile-time immediate (a swizzled patch cost ~150 VALU ops per 36 MFMAs and ran at 538 TFLOP/s).
#include "isp_common.h"

namespace {

typedef __attribute__((ext_vector_type(4))) short s16x4;

constexpr int WT = 8;                       // spatial tile: 8 x 8 output pixels
constexpr int WPP = 12;                     // patch pitch in pixels (10 used)
constexpr int WPATCH_PX = 10 * WPP;         // 120 patch pixels
constexpr int WC = 64, WN = 128;            // channels per block
constexpr int XPITCH = 160;                 // bytes per patch pixel: 128 data + 32 pad (10 chunks of 16 B)
constexpr int X_PIECES = (WPATCH_PX * 10 + 63) / 64;  // 19 one-KiB DMA pieces
constexpr int G_BYTES = WT * WT * WN * 2;   // 16 KiB
constexpr int X_BYTES = X_PIECES * 1024;    // 19 KiB
constexpr int W_STAGE = G_BYTES + X_BYTES;
constexpr int W_LDS = 2 * W_STAGE;

__device__ __forceinline__ int gswz(int row, int chunk) { return chunk ^ (((row & 3) << 1) | (((row >> 3) & 1) << 3)); }

__device__ __forceinline__ s16x4 tr_read(const char* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((ISP_LDS s16x4*)p);
}

__global__ __launch_bounds__(512) void conv3x3_wgrad_kernel(const bf16_t* __restrict__ g, const bf16_t* __restrict__ x,
                                                            float* __restrict__ dw, int B, int H, int W, int C, int N,
                                                            int tiles_y, int tiles_x, long tiles_per_chunk, int tiles_c,
                                                            int tiles_n) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    int bid = blockIdx.x;
    const int tc = bid % tiles_c;
    bid /= tiles_c;
    const int tn = bid % tiles_n;
    const long chunk = bid / tiles_n;
    const int c0 = tc * WC, n0 = tn * WN;
    const long ntiles = (long)B * tiles_y * tiles_x;
    const long t_begin = chunk * tiles_per_chunk;
    long t_end = t_begin + tiles_per_chunk;
    if (t_end > ntiles) t_end = ntiles;
    if (t_begin >= t_end) return;

    // ---- DMA slots.  g image: 16 pieces of 4 pixels x 256 B (waves take pieces w, w+8);
    //      patch image: 19 pieces of 64 chunks, chunk index -> (pixel = idx / 10, chunk = idx % 10; chunks 8, 9 = pad).
    int g_ty[2], g_tx[2], g_col[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int px = (wid + 8 * i) * 4 + (lane >> 4);
        g_ty[i] = px >> 3;
        g_tx[i] = px & 7;
        g_col[i] = gswz(px, lane & 15) * 8;
    }
    int x_py[3], x_px[3], x_col[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int idx = (wid + 8 * i) * 64 + lane;
        const int px = idx / 10, ch = idx - px * 10;
        x_py[i] = px / WPP;
        x_px[i] = ch < 8 && px < WPATCH_PX ? px - x_py[i] * WPP : WPP;  // WPP = never valid (pad chunk / tail)
        x_col[i] = ch * 8;
    }
    const bf16_t* const zero = reinterpret_cast<const bf16_t*>(g_isp_zero16);
    auto stage = [&](long t, char* buf) {
        const int b = (int)(t / ((long)tiles_y * tiles_x));
        const int rem = (int)(t - (long)b * tiles_y * tiles_x);
        const int y0 = (rem / tiles_x) * WT, x0 = (rem % tiles_x) * WT;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int y = y0 + g_ty[i], xx = x0 + g_tx[i];
            const bool ok = y < H && xx < W && n0 + g_col[i] < N;
            glds16(ok ? g + (((size_t)b * H + y) * W + xx) * N + n0 + g_col[i] : zero, buf + (wid + 8 * i) * 1024);
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            if (wid + 8 * i >= X_PIECES) continue;
            const int y = y0 - 1 + x_py[i], xx = x0 - 1 + x_px[i];
            const bool ok = x_px[i] < 10 && (unsigned)y < (unsigned)H && (unsigned)xx < (unsigned)W && c0 + x_col[i] < C;
            glds16(ok ? x + (((size_t)b * H + y) * W + xx) * C + c0 + x_col[i] : zero, buf + G_BYTES + (wid + 8 * i) * 1024);
        }
    };

    // ---- fragment geometry.  16-lane group fq of a transposed read addresses 4 consecutive pixels (q = fr>>2)
    //      x 16 channels (4p .. 4p+3, p = fr&3) and hands lane fr channel fr of those 4 pixels.
    const int fr = lane & 15, fq = lane >> 4, q = fr >> 2, p = fr & 3;
    const int cw = wid & 3, nh = wid >> 2;  // wave: input channels 16*cw .. +15, output channels 64*nh .. +63
    // g: row (pixel) = (4ks + fq)*8 + 4j + q ; the swizzle term depends on (row&3) = q and (row>>3)&1 = fq&1 only
    int g_off[4];
    {
        const int sw = (q << 1) | ((fq & 1) << 3);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const int col = nh * 64 + nt * 16 + 4 * p;
            g_off[nt] = (fq * 8 + q) * 256 + (((col >> 3) ^ sw) << 4) + (col & 7) * 2;
        }
    }
    // patch: pixel (4ks + fq + 1 + dy)*12 + 1 + dx + 4j + q = (fq*12 + q) + an immediate; channels 16cw + 4p ..
    const int x_off = (fq * WPP + q) * XPITCH + (cw * 16 + 4 * p) * 2;

    f32x4 acc[9][4];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[t][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

    stage(t_begin, smem);
    __syncthreads();
    for (long t = t_begin; t < t_end; ++t) {
        const char* buf = smem + ((t - t_begin) & 1) * W_STAGE;
        if (t + 1 < t_end) stage(t + 1, smem + (((t - t_begin) + 1) & 1) * W_STAGE);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 gf[4];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const s16x4 lo = tr_read(buf + ks * 8192 + g_off[nt]);
                const s16x4 hi = tr_read(buf + ks * 8192 + 1024 + g_off[nt]);
                gf[nt] = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            }
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int dy = tap / 3 - 1, dx = tap % 3 - 1;
                const int imm = ((4 * ks + 1 + dy) * WPP + 1 + dx) * XPITCH;  // compile-time
                const char* xb = buf + G_BYTES + x_off;
                const s16x4 lo = tr_read(xb + imm);                 // tile columns 0..3
                const s16x4 hi = tr_read(xb + imm + 4 * XPITCH);    // tile columns 4..7
                const bf16x8 xf = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
                    acc[tap][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf, gf[nt], acc[tap][nt], 0, 0, 0);
            }
        }
        __syncthreads();
    }

    // D[i = c][j = n]: lane holds c = c0 + 16cw + 4fq + r (r = 0..3), n = n0 + 64nh + 16nt + fr
    const long K9 = 9L * C;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        const int n = n0 + nh * 64 + nt * 16 + fr;
        if (n >= N) continue;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int c = c0 + cw * 16 + 4 * fq;
            float* o = dw + (size_t)n * K9 + (size_t)tap * C + c;
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (c + r < C) atomicAdd(o + r, acc[tap][nt][r]);
        }
    }
}

}  // namespace

extern "C" int isp_conv3x3_wgrad_bf16_atomic(const void* g, const void* x, float* dw, int B, int H, int W, int C, int N,
                                             void* stream) {
    ISP_CHECK_ARG(g && x && dw && B > 0 && H > 0 && W > 0 && C > 0 && N > 0 && C % 8 == 0 && N % 8 == 0);
    const int tiles_y = (H + WT - 1) / WT, tiles_x = (W + WT - 1) / WT;
    const int tiles_c = (C + WC - 1) / WC, tiles_n = (N + WN - 1) / WN;
    const long ntiles = (long)B * tiles_y * tiles_x;
    // split-K over pixel tiles: one round of blocks on 256 CUs (1 block/CU; every block ends with 73 k fp32 atomics),
    // at least 8 tiles per block
    long chunks = 256 / ((long)tiles_c * tiles_n);
    if (chunks > (ntiles + 7) / 8) chunks = (ntiles + 7) / 8;
    if (chunks < 1) chunks = 1;
    const long per = (ntiles + chunks - 1) / chunks;
    chunks = (ntiles + per - 1) / per;
    const long nwg = chunks * tiles_c * tiles_n;
    ISP_CHECK_ARG(nwg <= 0x7fffffffL);
    static bool attr_done = false;
    if (!attr_done) {
        if (hipFuncSetAttribute((const void*)conv3x3_wgrad_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, W_LDS) !=
            hipSuccess)
            return ISP_ERR_LAUNCH;
        attr_done = true;
    }
    conv3x3_wgrad_kernel<<<(unsigned)nwg, 512, W_LDS, (hipStream_t)stream>>>(
        (const bf16_t*)g, (const bf16_t*)x, dw, B, H, W, C, N, tiles_y, tiles_x, per, tiles_c, tiles_n);
    return isp_launch_status();
}
