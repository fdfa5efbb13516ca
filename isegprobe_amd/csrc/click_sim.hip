// Train-time click simulation on the device: get_next_points (reference core/training/trainer.py:575-618).
//
// For every sample: FN / FP masks of the current prediction, cv2.distanceTransform(mask, DIST_L2, 5) of the
// 1-pixel zero-padded masks (OpenCV's two-pass 5x5 chamfer transform in 16.16 fixed point, step costs 1 / 1.4 /
// 2.1969), the larger maximum picks the click polarity, the click is drawn uniformly among the pixels whose
// distance exceeds half that maximum and written into the points tensor.  The reference does this on the host
// (two device->host copies and 2*B sequential OpenCV calls per simulated click, up to 3 clicks per step); here the
// probabilities never leave HBM and nothing synchronises with the host.
//
// The chamfer passes are raster scans with a causal 5x5 half-neighbourhood.  Pixel (i, j) depends on (i, j-1),
// (i-1, j-2..j+2) and (i-2, j-1), (i-2, j+1): with time t = j + 3i all dependencies have a smaller t, so one thread
// per image row sweeps its row and all rows advance together, one column per barrier (wavefront).  The few
// neighbouring values a row needs from the two rows above travel through a small LDS ring (a row's last eight values); the
// full integer plane goes to the workspace -- skewed, one plane row per sweep step -- for the backward pass (mirror image of
// the forward pass) and the selection.
// The uniform draw is explicit: rand32[b] (a 32-bit integer per sample supplied by the caller) -> index
// (rand32 * n) >> 32 in row-major order, where the reference calls np.random.randint(0, n).
#include "isp_common.h"

namespace {

constexpr unsigned INIT_DIST0 = 0x7fffffffu >> 2;
constexpr unsigned HV = 65536u, DIAG = 91750u, LONGD = 143976u;  // round(1, 1.4f, 2.1969f * 2^16)
constexpr int RING = 8;  // a row's last 8 values: rows i-1 / i-2 are 3 / 6 columns ahead and are read at most 2 / 1 columns back
constexpr int MAXROWS = 1024;

__device__ __forceinline__ unsigned umin(unsigned a, unsigned b) { return a < b ? a : b; }

// grid (2 masks, B), block = PH rounded up to 64 (<= 1024).  plane: [B][2][T][PHS] uint32 (final distances), SKEWED: padded
// pixel (i, j) lives at [j + 3 i][i], T = PW + 3 (PH - 1) sweep steps, PHS = PH rounded up to 64 -- at a sweep step every row
// (= lane) touches the same plane row t, so the per-step store / load of a wave is 256 contiguous bytes.  (Row-major planes made
// every step a 64-line scatter: with the barrier and the LDS ring fixed the kernel still took 0.6 us per step on the CU's
// vector-memory path alone.)  In the backward pass the forward value of step tb sits at plane row T - 1 - tb for every row.
// A sweep step is: seven ring reads, eight min/add pairs, one ring write, ONE barrier.  What a step must not contain is a wait
// for global memory: the mask is ballot-packed into LDS bits up front (the first version evaluated gt / pred per step: two
// global loads in the dependency chain), the forward plane goes out with fire-and-forget stores and comes back in the backward
// pass through register prefetch eight steps ahead, and the barrier is a raw s_barrier behind an LDS-only wait (__syncthreads
// would also drain the stores and the prefetch: that alone made a step ~0.6 us -- 1.1 ms per 224^2 launch, 2.9 ms at 448^2).
// One barrier per step suffices: the slot row i writes at step t (column j, slot j & 7 = the slot of column j - 8) is read in
// the same step by nobody (row i+1 reads columns j-5 .. j-1 of row i, row i+2 columns j-7 and j-5), its previous content
// (column j - 8) was last read by row i+2 at step t - 1, and the value is needed from step t + 1 on.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
}

__global__ __launch_bounds__(1024) void chamfer5_kernel(const float* __restrict__ pred, const float* __restrict__ gt,
                                                         unsigned* __restrict__ planes, float* __restrict__ maxima,
                                                         int H, int W, float thr) {
    // a row's last 8 values, slot-major ([slot][row + 2]: the lanes of a wave = consecutive rows hit consecutive banks; row-major
    // [row][8] put lanes 8 rows apart on one bank, 8-way conflicts on every access)
    __shared__ unsigned ring[RING][MAXROWS + 4];
    __shared__ unsigned wave_max[16];
    extern __shared__ unsigned mask_bits[];  // [H][WPR]: bit c of row r = mask value at interior pixel (r, c)
    const int m = blockIdx.x, b = blockIdx.y;
    const int PH = H + 2, PW = W + 2;
    const int WPR = 2 * ((W + 63) / 64);
    const int i = threadIdx.x;  // padded row
    const bool live = i < PH;
    const int steps = PW + 3 * (PH - 1), PHS = (PH + 63) & ~63;
    unsigned* plane = planes + ((size_t)b * 2 + m) * steps * PHS;
    const float* pr = pred + (size_t)b * H * W;
    const float* gr = gt + (size_t)b * H * W;
    // ring rows are offset by 2: rows -2, -1 (above the padded image) stay INIT_DIST0
    for (int k = threadIdx.x; k < (MAXROWS + 4) * RING; k += blockDim.x) (&ring[0][0])[k] = INIT_DIST0;
    {   // the mask, 64 columns per wave and ballot (coalesced reads of gt / pred, once)
        const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nwv = blockDim.x >> 6, segs = WPR / 2;
        for (int it = wv; it < H * segs; it += nwv) {
            const int r = it / segs, sgm = it - r * segs, c = sgm * 64 + lane;
            bool in = false;
            if (c < W) {
                const size_t idx = (size_t)r * W + c;
                const bool g = gr[idx] > 0.5f;
                const float p = pr[idx];
                in = m == 0 ? (g && p < thr) : (!g && p > thr);
            }
            const unsigned long long bal = __ballot(in);
            if (lane == 0) {
                mask_bits[r * WPR + 2 * sgm] = (unsigned)bal;
                mask_bits[r * WPR + 2 * sgm + 1] = (unsigned)(bal >> 32);
            }
        }
    }
    __syncthreads();
    const bool row_in = i >= 1 && i <= H;
    const unsigned* const rb = mask_bits + (row_in ? i - 1 : 0) * WPR;
    int cw = -1;
    unsigned wbits = 0;
    auto inside = [&](int j) -> bool {  // mask value at padded (i, j)
        if (!row_in || j < 1 || j > W) return false;
        const int c = j - 1;
        if ((c >> 5) != cw) {
            cw = c >> 5;
            wbits = rb[cw];
        }
        return (wbits >> (c & 31)) & 1u;
    };

    // ---- forward pass: t = j + 3i.  Every row sweeps two VIRTUAL columns PW, PW + 1 behind its last one and writes the "outside"
    // value into their ring slots, and column -1 finds it in slot 7 (not written before column 7): the neighbour reads need no
    // bounds checks (six compare + select pairs per step: the kernel is bound by its instruction count by now).
    unsigned left = INIT_DIST0;
    {
        int j = -3 * i;
        unsigned* pp = plane + i;  // plane row t, this row's element
        for (int t = 0; t < steps + 2; ++t, ++j, pp += PHS) {
            if (live && (unsigned)j < (unsigned)(PW + 2)) {
                unsigned v = INIT_DIST0;
                if (j < PW) {
                    v = 0;
                    if (inside(j)) {
                        const int r1 = i + 1, r2 = i;  // ring rows of image rows i-1, i-2
                        // all seven reads first, unconditionally.  (Written as `oob ? INIT : ring[..]` each read became its own
                        // branch with its own lgkmcnt(0) wait: seven LDS round trips in series per step, ~1 400 cycles -- the
                        // whole cost of the first version of this kernel.)
                        const unsigned a0 = ring[(j - 1) & (RING - 1)][r2], a1 = ring[(j + 1) & (RING - 1)][r2];
                        const unsigned b0 = ring[(j - 2) & (RING - 1)][r1], b1 = ring[(j - 1) & (RING - 1)][r1], b2 = ring[j & (RING - 1)][r1];
                        const unsigned b3 = ring[(j + 1) & (RING - 1)][r1], b4 = ring[(j + 2) & (RING - 1)][r1];
                        unsigned t0 = umin(a0, a1) + LONGD;
                        t0 = umin(t0, umin(b0, b4) + LONGD);
                        t0 = umin(t0, umin(b1, b3) + DIAG);
                        t0 = umin(t0, umin(b2, left) + HV);
                        v = t0;
                    }
                    left = v;
                    *pp = v;
                }
                ring[j & (RING - 1)][i + 2] = v;
            }
            lds_barrier();
        }
    }

    // ---- backward pass: mirror image (rows below, columns to the right), t = (PW-1-j) + 3(PH-1-i)
    __syncthreads();  // (also completes this thread's plane stores before it reads them back)
    for (int k = threadIdx.x; k < (MAXROWS + 4) * RING; k += blockDim.x) (&ring[0][0])[k] = INIT_DIST0;
    __syncthreads();
    unsigned right = INIT_DIST0, row_max = 0;
    const int jb0 = PW - 1 + 3 * (PH - 1 - i);  // column at step t: jb0 - t
    unsigned curv[8], nxtv[8];
    auto fetch = [&](unsigned (&dst)[8], int t0) {  // forward values of the eight steps t0 .. t0 + 7 (0 where inactive)
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int j = jb0 - (t0 + u);
            dst[u] = (live && j >= 0 && j < PW) ? plane[(size_t)(steps - 1 - (t0 + u)) * PHS + i] : 0u;
        }
    };
    fetch(curv, 0);
    {
        unsigned* pp = plane + (size_t)(steps - 1) * PHS + i;  // plane row steps - 1 - t
        int j = jb0;
        const bool interior_row = i >= 1 && i <= H;
        for (int t0 = 0; t0 < steps + 2; t0 += 8) {  // (two virtual columns -1, -2 behind column 0, as in the forward pass)
            fetch(nxtv, t0 + 8);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (t0 + u < steps + 2) {  // (block-uniform)
                    if (live && (unsigned)(j + 2) < (unsigned)(PW + 2)) {
                        unsigned t0v = INIT_DIST0;
                        if (j >= 0) {
                            t0v = curv[u];
                            if (t0v > HV) {
                                const int r1 = i + 3, r2 = i + 4;  // ring rows of image rows i+1, i+2 (ring row index = row + 2)
                                const unsigned a0 = ring[(j + 1) & (RING - 1)][r2], a1 = ring[(j - 1) & (RING - 1)][r2];
                                const unsigned b0 = ring[(j + 2) & (RING - 1)][r1], b1 = ring[(j + 1) & (RING - 1)][r1], b2 = ring[j & (RING - 1)][r1];
                                const unsigned b3 = ring[(j - 1) & (RING - 1)][r1], b4 = ring[(j - 2) & (RING - 1)][r1];
                                t0v = umin(t0v, umin(a0, a1) + LONGD);
                                t0v = umin(t0v, umin(b0, b4) + LONGD);
                                t0v = umin(t0v, umin(b1, b3) + DIAG);
                                t0v = umin(t0v, umin(b2, right) + HV);
                            }
                            right = t0v;
                            *pp = t0v;
                            if (interior_row && j >= 1 && j <= W) row_max = t0v > row_max ? t0v : row_max;  // dt[1:-1, 1:-1]
                        }
                        ring[j & (RING - 1)][i + 2] = t0v;
                    }
                    lds_barrier();
                    --j;
                    pp -= PHS;
                }
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) curv[u] = nxtv[u];
        }
    }
    // ---- maximum over the interior (as float, the way cv2 returns it)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned other = __shfl_xor(row_max, o);
        row_max = other > row_max ? other : row_max;
    }
    if ((threadIdx.x & 63) == 0) wave_max[threadIdx.x >> 6] = row_max;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned mx = 0;
        for (int k = 0; k < (int)(blockDim.x >> 6); ++k) mx = wave_max[k] > mx ? wave_max[k] : mx;
        maxima[b * 2 + m] = (float)mx * (1.0f / 65536.0f);
    }
}

// grid B, block = H rounded up to 64: uniform draw among {dt > max/2} of the chosen polarity, row-major order.  Thread r owns
// interior row r and walks its columns in the skewed plane (pixel (r+1, c+1) at plane row c + 1 + 3 (r + 1)): at walk step q the
// lanes read one plane row, 256 contiguous bytes per wave.
__global__ __launch_bounds__(1024) void click_select_kernel(const unsigned* __restrict__ planes,
                                                             const float* __restrict__ maxima,
                                                             const unsigned* __restrict__ rand32, float* __restrict__ points,
                                                             int H, int W, int P, int click_indx) {
    __shared__ unsigned prefix[MAXROWS + 1];
    const int b = blockIdx.x, r = threadIdx.x;  // r: interior row
    const int PW = W + 2, PH = H + 2, steps = PW + 3 * (PH - 1), PHS = (PH + 63) & ~63;
    const float fn_max = maxima[b * 2], fp_max = maxima[b * 2 + 1];
    const bool positive = fn_max > fp_max;  // trainer.py:601
    const float half = fmaxf(fn_max, fp_max) / 2.0f;
    // interior pixel (r, c): plane[(c + 4 + 3 r) * PHS + r + 1]
    const unsigned* plane = planes + ((size_t)b * 2 + (positive ? 0 : 1)) * steps * PHS + (size_t)(4 + 3 * r) * PHS + r + 1;
    unsigned cnt = 0;
    if (r < H)
        for (int c = 0; c < W; ++c) cnt += ((float)plane[(size_t)c * PHS] * (1.0f / 65536.0f) > half);
    if (r <= MAXROWS) prefix[r] = r < H ? cnt : 0;
    __syncthreads();
    if (threadIdx.x == 0) {  // exclusive scan over <= 1024 rows
        unsigned run = 0;
        for (int k = 0; k < H; ++k) {
            const unsigned c = prefix[k];
            prefix[k] = run;
            run += c;
        }
        prefix[H] = run;
    }
    __syncthreads();
    const unsigned n = prefix[H];
    if (n == 0 || r >= H) return;  // trainer.py:605: no inner pixel -> points unchanged
    const unsigned k = (unsigned)(((unsigned long long)rand32[b] * n) >> 32);
    if (k < prefix[r] || k >= prefix[r] + cnt) return;
    unsigned seen = prefix[r];
    for (int c = 0; c < W; ++c) {
        if ((float)plane[(size_t)c * PHS] * (1.0f / 65536.0f) > half) {
            if (seen == k) {
                float* p = points + ((size_t)b * 2 * P + (positive ? P : 2 * P) - click_indx) * 3;
                p[0] = (float)r, p[1] = (float)c, p[2] = (float)click_indx;
                return;
            }
            ++seen;
        }
    }
}

}  // namespace

static long plane_words(int H, int W) { return (long)(W + 2 + 3 * (H + 1)) * ((H + 2 + 63) & ~63); }

extern "C" long isp_next_points_workspace_bytes(int B, int H, int W) {
    if (B <= 0 || H <= 0 || W <= 0) return ISP_ERR_INVALID;
    return plane_words(H, W) * 4 * 2 * B + (long)B * 2 * 4 + 64;  // skewed planes (see chamfer5_kernel), two masks per sample
}

extern "C" int isp_next_points(const float* pred, const float* gt, float* points, const unsigned* rand32, int B, int H,
                               int W, int P, int click_indx, float pred_thresh, void* workspace, void* stream) {
    ISP_CHECK_ARG(pred && gt && points && rand32 && workspace && B > 0 && H > 0 && W > 0 && P > 0);
    ISP_CHECK_ARG(click_indx > 0 && click_indx <= P && H + 2 <= MAXROWS && B <= 65535);
    ISP_CHECK_ARG((long)(H + W) * 143976L < (long)INIT_DIST0);  // distances stay below the "outside" marker
    hipStream_t s = (hipStream_t)stream;
    unsigned* planes = (unsigned*)workspace;
    float* maxima = (float*)((char*)workspace + (size_t)B * 2 * plane_words(H, W) * 4);
    const int t1 = ((H + 2 + 63) / 64) * 64, t2 = ((H + 63) / 64) * 64;
    const int mask_bytes = H * 2 * ((W + 63) / 64) * 4;  // ballot-packed mask, [H][2 ceil(W / 64)] words
    ISP_CHECK_ARG(mask_bytes <= 120 * 1024);            // + 33 KiB of ring: up to 960^2
    static bool attr_done = false;
    if (!attr_done) {
        if (hipFuncSetAttribute((const void*)chamfer5_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024) != hipSuccess)
            return ISP_ERR_LAUNCH;
        attr_done = true;
    }
    chamfer5_kernel<<<dim3(2, B), t1, mask_bytes, s>>>(pred, gt, planes, maxima, H, W, pred_thresh);
    click_select_kernel<<<B, t2, 0, s>>>(planes, maxima, rand32, points, H, W, P, click_indx);
    return isp_launch_status();
}
