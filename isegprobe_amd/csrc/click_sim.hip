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
// neighbouring values a row needs from the two rows above travel through a small per-row LDS ring; the full
// integer plane goes to the workspace for the backward pass (mirror image of the forward pass) and the selection.
// The uniform draw is explicit: rand32[b] (a 32-bit integer per sample supplied by the caller) -> index
// (rand32 * n) >> 32 in row-major order, where the reference calls np.random.randint(0, n).
#include "isp_common.h"

namespace {

constexpr unsigned INIT_DIST0 = 0x7fffffffu >> 2;
constexpr unsigned HV = 65536u, DIAG = 91750u, LONGD = 143976u;  // round(1, 1.4f, 2.1969f * 2^16)
constexpr int RING = 8;  // a row's last 8 values: rows i-1 / i-2 are 3 / 6 columns ahead and are read at most 2 / 1 columns back
constexpr int MAXROWS = 1024;

__device__ __forceinline__ unsigned umin(unsigned a, unsigned b) { return a < b ? a : b; }

// grid (2 masks, B), block = PH rounded up to 64 (<= 1024).  plane: [B][2][PH][PW] uint32 (final distances).
__global__ __launch_bounds__(1024) void chamfer5_kernel(const float* __restrict__ pred, const float* __restrict__ gt,
                                                         unsigned* __restrict__ planes, float* __restrict__ maxima,
                                                         int H, int W, float thr) {
    __shared__ unsigned ring[MAXROWS + 4][RING];
    __shared__ unsigned wave_max[16];
    const int m = blockIdx.x, b = blockIdx.y;
    const int PH = H + 2, PW = W + 2;
    const int i = threadIdx.x;  // padded row
    const bool live = i < PH;
    unsigned* plane = planes + ((size_t)b * 2 + m) * PH * PW;
    const float* pr = pred + (size_t)b * H * W;
    const float* gr = gt + (size_t)b * H * W;
    // ring rows are offset by 2: rows -2, -1 (above the padded image) stay INIT_DIST0
    for (int k = threadIdx.x; k < (MAXROWS + 4) * RING; k += blockDim.x) (&ring[0][0])[k] = INIT_DIST0;
    __syncthreads();

    auto inside = [&](int j) -> bool {  // mask value at padded (i, j)
        if (i < 1 || i > H || j < 1 || j > W) return false;
        const size_t idx = (size_t)(i - 1) * W + (j - 1);
        const bool g = gr[idx] > 0.5f;
        const float p = pr[idx];
        return m == 0 ? (g && p < thr) : (!g && p > thr);
    };

    // ---- forward pass: t = j + 3i
    unsigned left = INIT_DIST0;
    const int steps = PW + 3 * (PH - 1);
    for (int t = 0; t < steps; ++t) {
        const int j = t - 3 * i;
        unsigned v = 0;
        const bool act = live && j >= 0 && j < PW;
        if (act) {
            if (inside(j)) {
                const unsigned* r1 = ring[i + 1];  // row i-1
                const unsigned* r2 = ring[i];      // row i-2
                auto at = [&](const unsigned* r, int c) { return (c < 0 || c >= PW) ? INIT_DIST0 : r[c & (RING - 1)]; };
                unsigned t0 = at(r2, j - 1) + LONGD;
                t0 = umin(t0, at(r2, j + 1) + LONGD);
                t0 = umin(t0, at(r1, j - 2) + LONGD);
                t0 = umin(t0, at(r1, j - 1) + DIAG);
                t0 = umin(t0, at(r1, j) + HV);
                t0 = umin(t0, at(r1, j + 1) + DIAG);
                t0 = umin(t0, at(r1, j + 2) + LONGD);
                t0 = umin(t0, left + HV);
                v = t0;
            }
            left = v;
            plane[(size_t)i * PW + j] = v;
        }
        __syncthreads();  // everyone has read this step's neighbours
        if (act) ring[i + 2][j & (RING - 1)] = v;
        __syncthreads();
    }

    // ---- backward pass: mirror image (rows below, columns to the right), t = (PW-1-j) + 3(PH-1-i)
    for (int k = threadIdx.x; k < (MAXROWS + 4) * RING; k += blockDim.x) (&ring[0][0])[k] = INIT_DIST0;
    __syncthreads();
    unsigned right = INIT_DIST0, row_max = 0;
    for (int t = 0; t < steps; ++t) {
        const int j = PW - 1 - (t - 3 * (PH - 1 - i));
        unsigned v = 0;
        const bool act = live && j >= 0 && j < PW;
        if (act) {
            unsigned t0 = plane[(size_t)i * PW + j];
            if (t0 > HV) {
                const unsigned* r1 = ring[i + 3];  // row i+1 (ring row index = row + 2)
                const unsigned* r2 = ring[i + 4];  // row i+2
                auto at = [&](const unsigned* r, int c) { return (c < 0 || c >= PW) ? INIT_DIST0 : r[c & (RING - 1)]; };
                t0 = umin(t0, at(r2, j + 1) + LONGD);
                t0 = umin(t0, at(r2, j - 1) + LONGD);
                t0 = umin(t0, at(r1, j + 2) + LONGD);
                t0 = umin(t0, at(r1, j + 1) + DIAG);
                t0 = umin(t0, at(r1, j) + HV);
                t0 = umin(t0, at(r1, j - 1) + DIAG);
                t0 = umin(t0, at(r1, j - 2) + LONGD);
                t0 = umin(t0, right + HV);
            }
            v = t0;
            right = v;
            plane[(size_t)i * PW + j] = v;
            if (i >= 1 && i <= H && j >= 1 && j <= W) row_max = v > row_max ? v : row_max;  // dt[1:-1, 1:-1]
        }
        __syncthreads();
        if (act) ring[i + 2][j & (RING - 1)] = v;
        __syncthreads();
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

// grid B, block = H rounded up to 64: uniform draw among {dt > max/2} of the chosen polarity, row-major order
__global__ __launch_bounds__(1024) void click_select_kernel(const unsigned* __restrict__ planes,
                                                             const float* __restrict__ maxima,
                                                             const unsigned* __restrict__ rand32, float* __restrict__ points,
                                                             int H, int W, int P, int click_indx) {
    __shared__ unsigned prefix[MAXROWS + 1];
    const int b = blockIdx.x, r = threadIdx.x;  // r: interior row
    const int PW = W + 2, PH = H + 2;
    const float fn_max = maxima[b * 2], fp_max = maxima[b * 2 + 1];
    const bool positive = fn_max > fp_max;  // trainer.py:601
    const float half = fmaxf(fn_max, fp_max) / 2.0f;
    const unsigned* plane = planes + ((size_t)b * 2 + (positive ? 0 : 1)) * PH * PW + (size_t)(r + 1) * PW + 1;
    unsigned cnt = 0;
    if (r < H)
        for (int c = 0; c < W; ++c) cnt += ((float)plane[c] * (1.0f / 65536.0f) > half);
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
        if ((float)plane[c] * (1.0f / 65536.0f) > half) {
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

extern "C" long isp_next_points_workspace_bytes(int B, int H, int W) {
    if (B <= 0 || H <= 0 || W <= 0) return ISP_ERR_INVALID;
    return (long)B * 2 * (H + 2) * (W + 2) * 4 + (long)B * 2 * 4 + 64;
}

extern "C" int isp_next_points(const float* pred, const float* gt, float* points, const unsigned* rand32, int B, int H,
                               int W, int P, int click_indx, float pred_thresh, void* workspace, void* stream) {
    ISP_CHECK_ARG(pred && gt && points && rand32 && workspace && B > 0 && H > 0 && W > 0 && P > 0);
    ISP_CHECK_ARG(click_indx > 0 && click_indx <= P && H + 2 <= MAXROWS && B <= 65535);
    ISP_CHECK_ARG((long)(H + W) * 143976L < (long)INIT_DIST0);  // distances stay below the "outside" marker
    hipStream_t s = (hipStream_t)stream;
    unsigned* planes = (unsigned*)workspace;
    float* maxima = (float*)((char*)workspace + (size_t)B * 2 * (H + 2) * (W + 2) * 4);
    const int t1 = ((H + 2 + 63) / 64) * 64, t2 = ((H + 63) / 64) * 64;
    chamfer5_kernel<<<dim3(2, B), t1, 0, s>>>(pred, gt, planes, maxima, H, W, pred_thresh);
    click_select_kernel<<<B, t2, 0, s>>>(planes, maxima, rand32, points, H, W, P, click_indx);
    return isp_launch_status();
}
