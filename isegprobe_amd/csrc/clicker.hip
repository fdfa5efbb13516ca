// Device-side robot user: next click + IoU of one prediction, without leaving the GPU.
//
// Replaces, per click of the NoC loop, Clicker._get_next_click (reference core/inference/clicker.py:58-91:
// FN / FP masks, exact Euclidean distance transform of the 1-pixel zero-padded masks via
// cv2.distanceTransform(DIST_L2, maskSize 0), zeroing of already-clicked pixels, "larger maximum wins",
// first maximum in row-major order) and utils.get_iou (core/inference/utils.py:107-120), which the
// reference runs on the host after two device->host copies of the probability map (base_predictor.py:108,
// evaluation.py:73-76).
//
// Everything is integer: the squared EDT is exact in int32, sqrt is monotonic, so comparing squared
// distances gives the same click as comparing OpenCV's float32 distances.
//   1. columns: g[m][y][x] = vertical distance from (y,x) to the nearest zero of mask m in column x,
//      with virtual zeros at y = -1 and y = H (the reference's 1-pixel padding); IoU counts on the way.
//   2. rows:    d2(y,x) = min_x' (x-x')^2 + g[m][y][x']^2 with virtual zeros at x' = -1 and x' = W; the
//      scan walks outwards from x and stops once (x-x')^2 >= best.  Masked by not_clicked, packed as
//      (d2 << 32 | ~index) and max-reduced: largest distance, then smallest row-major index.
//   3. decide:  positive click iff max(FN) > max(FP) (clicker.py:84), write the result record.
#include "isp_common.h"

namespace {

constexpr int MAXW = 8192;

__device__ __forceinline__ bool in_mask(int m, unsigned char pred, unsigned char gt, unsigned char ni) {
    return m == 0 ? (gt && !pred && ni) : (!gt && pred && ni);  // 0: false negatives, 1: false positives
}

// Columns, H <= 1024: a column is cut into <= 16 segments of 32 or 64 rows, one thread per (column, segment).  A thread loads its
// rows once, keeps "this pixel is a zero of mask m" as a 64-bit word per mask, publishes first / last zero of its segment in LDS,
// finds the nearest zero above / below its segment in the other segments' summaries, and every row's two vertical distances are
// bit scans of the word -- no serial chain over the column, one round of loads.  (The one-thread-per-column scan below walks 2 H
// rows in 16-row batches and runs 10 waves on the whole GPU for a 480 x 640 image: 287 us per click, 10 % of a 448^2 click of
// the bilinear probe.)
__global__ __launch_bounds__(1024) void clicker_columns_seg_kernel(const unsigned char* __restrict__ pred,
                                                                   const unsigned char* __restrict__ gt,
                                                                   const unsigned char* __restrict__ ni, int* __restrict__ g,
                                                                   int* __restrict__ counts, int H, int W, int SEG) {
    __shared__ short s_first[2][16][64], s_last[2][16][64];
    const int xl = threadIdx.x & 63, sg = threadIdx.x >> 6, nseg = blockDim.x >> 6;
    const int x = blockIdx.x * 64 + xl;
    const int y0 = sg * SEG;
    const bool col = x < W;
    unsigned long long z0 = 0, z1 = 0;  // bit k: row y0 + k is a zero of mask 0 / 1 (rows >= H: the virtual zero row)
    int inter = 0, uni = 0;
    if (col) {
        for (int kb = 0; kb < SEG; kb += 16) {
            unsigned char p[16], t[16], n[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int y = y0 + kb + k < H ? y0 + kb + k : H - 1;
                const size_t i = (size_t)y * W + x;
                p[k] = pred[i], t[k] = gt[i], n[k] = ni ? ni[i] : 1;
            }
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const bool real = y0 + kb + k < H;
                const unsigned long long bit = 1ull << (kb + k);
                if (!real || !in_mask(0, p[k], t[k], n[k])) z0 |= bit;
                if (!real || !in_mask(1, p[k], t[k], n[k])) z1 |= bit;
                inter += real && (p[k] && t[k] && n[k]);
                uni += real && ((p[k] || t[k]) && n[k]);
            }
        }
    }
    if (SEG < 64) z0 &= (1ull << SEG) - 1, z1 &= (1ull << SEG) - 1;
    s_first[0][sg][xl] = z0 ? (short)(__ffsll((long long)z0) - 1) : (short)64;
    s_first[1][sg][xl] = z1 ? (short)(__ffsll((long long)z1) - 1) : (short)64;
    s_last[0][sg][xl] = z0 ? (short)(63 - __clzll((long long)z0)) : (short)-1;
    s_last[1][sg][xl] = z1 ? (short)(63 - __clzll((long long)z1)) : (short)-1;
    __syncthreads();
    if (col && y0 < H) {
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const unsigned long long z = m ? z1 : z0;
            int lz_in = -1, nz_in = H;  // nearest zero above / below the segment (virtual zero rows -1 and H)
            for (int q = sg - 1; q >= 0; --q)
                if (s_last[m][q][xl] >= 0) {
                    lz_in = q * SEG + s_last[m][q][xl];
                    break;
                }
            for (int q = sg + 1; q < nseg; ++q)
                if (s_first[m][q][xl] < 64) {
                    nz_in = q * SEG + s_first[m][q][xl];
                    break;
                }
            nz_in = nz_in < H ? nz_in : H;
            int* gm = g + (size_t)m * H * W + x;
            const int rows = H - y0 < SEG ? H - y0 : SEG;
            for (int k = 0; k < rows; ++k) {
                const int y = y0 + k;
                const unsigned long long below = z & ((2ull << k) - 1), above = z >> k;
                const int lz = below ? y0 + 63 - __clzll((long long)below) : lz_in;
                int nz = above ? y + __ffsll((long long)above) - 1 : nz_in;
                nz = nz < H ? nz : H;
                const int up = y - lz, dn = nz - y;
                gm[(size_t)y * W] = up < dn ? up : dn;
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) inter += __shfl_xor(inter, o), uni += __shfl_xor(uni, o);
    if (xl == 0 && (inter | uni)) {
        atomicAdd(counts, inter);
        atomicAdd(counts + 1, uni);
    }
}

// Taller images: one thread per column, both masks at once.  The scan is a serial chain over rows, so the byte loads of
// CH rows are issued together into registers before the chain consumes them (otherwise every row costs one
// full memory latency: 447 us for 480 x 640 before, vs the ~10 us the traffic needs).
__global__ __launch_bounds__(64) void clicker_columns_kernel(const unsigned char* __restrict__ pred,
                                                             const unsigned char* __restrict__ gt,
                                                             const unsigned char* __restrict__ ni, int* __restrict__ g,
                                                             int* __restrict__ counts, int H, int W) {
    constexpr int CH = 16;
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= W) return;
    int* g0 = g;                      // false negatives
    int* g1 = g + (size_t)H * W;      // false positives
    int d0 = 0, d1 = 0, inter = 0, uni = 0;
    for (int yb = 0; yb < H; yb += CH) {  // distance to the nearest zero above (virtual zero row at y = -1)
        unsigned char p[CH], t[CH], n[CH];
#pragma unroll
        for (int k = 0; k < CH; ++k) {
            const int y = yb + k < H ? yb + k : H - 1;
            const size_t i = (size_t)y * W + x;
            p[k] = pred[i], t[k] = gt[i], n[k] = ni ? ni[i] : 1;
        }
#pragma unroll
        for (int k = 0; k < CH; ++k) {
            if (yb + k >= H) break;
            const size_t i = (size_t)(yb + k) * W + x;
            d0 = in_mask(0, p[k], t[k], n[k]) ? d0 + 1 : 0;
            d1 = in_mask(1, p[k], t[k], n[k]) ? d1 + 1 : 0;
            g0[i] = d0;
            g1[i] = d1;
            inter += (p[k] && t[k] && n[k]);
            uni += ((p[k] || t[k]) && n[k]);
        }
    }
    d0 = d1 = 0;
    for (int yb = H - 1; yb >= 0; yb -= CH) {  // ... and below (virtual zero row at y = H)
        int u0[CH], u1[CH];
#pragma unroll
        for (int k = 0; k < CH; ++k) {
            const int y = yb - k >= 0 ? yb - k : 0;
            const size_t i = (size_t)y * W + x;
            u0[k] = g0[i], u1[k] = g1[i];
        }
#pragma unroll
        for (int k = 0; k < CH; ++k) {
            if (yb - k < 0) break;
            const size_t i = (size_t)(yb - k) * W + x;
            d0 = u0[k] ? d0 + 1 : 0;
            d1 = u1[k] ? d1 + 1 : 0;
            g0[i] = u0[k] < d0 ? u0[k] : d0;
            g1[i] = u1[k] < d1 ? u1[k] : d1;
        }
    }
    atomicAdd(counts, inter);
    atomicAdd(counts + 1, uni);
}

__global__ __launch_bounds__(256) void clicker_rows_kernel(const int* __restrict__ g,
                                                           const unsigned char* __restrict__ not_clicked,
                                                           unsigned long long* __restrict__ keys, int H, int W) {
    __shared__ int g2[MAXW];
    __shared__ unsigned long long best_key[4];
    const int y = blockIdx.x, m = blockIdx.y;
    const int* gr = g + ((size_t)m * H + y) * W;
    for (int x = threadIdx.x; x < W; x += blockDim.x) {
        const int v = gr[x];
        g2[x] = v * v;
    }
    __syncthreads();
    unsigned long long key = 0;
    for (int x = threadIdx.x; x < W; x += blockDim.x) {
        int best = g2[x];
        if (best > 0) {
            // virtual zero columns at -1 and W bound the search radius
            const int lb = (x + 1) * (x + 1), rb = (W - x) * (W - x);
            best = best < lb ? best : lb;
            best = best < rb ? best : rb;
            for (int r = 1; r * r < best; ++r) {
                if (x - r >= 0) {
                    const int c = r * r + g2[x - r];
                    best = c < best ? c : best;
                }
                if (x + r < W) {
                    const int c = r * r + g2[x + r];
                    best = c < best ? c : best;
                }
            }
        }
        const unsigned idx = (unsigned)(y * W + x);
        const unsigned d2 = not_clicked[idx] ? (unsigned)best : 0u;
        const unsigned long long k = ((unsigned long long)d2 << 32) | (unsigned long long)(0xffffffffu - idx);
        key = k > key ? k : key;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long other = __shfl_xor(key, o);
        key = other > key ? other : key;
    }
    if ((threadIdx.x & 63) == 0) best_key[threadIdx.x >> 6] = key;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < (int)(blockDim.x >> 6); ++i) key = best_key[i] > key ? best_key[i] : key;
        atomicMax(keys + m, key);
    }
}

__global__ void clicker_decide_kernel(const unsigned long long* __restrict__ keys, const int* __restrict__ counts, int W,
                                      int* __restrict__ out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const unsigned fn_d2 = (unsigned)(keys[0] >> 32), fp_d2 = (unsigned)(keys[1] >> 32);
    const int positive = fn_d2 > fp_d2;  // clicker.py:84 (ties -> negative click)
    const unsigned idx = 0xffffffffu - (unsigned)(keys[positive ? 0 : 1] & 0xffffffffu);
    out[0] = positive;
    out[1] = (int)(idx / (unsigned)W);
    out[2] = (int)(idx % (unsigned)W);
    out[3] = (int)fn_d2;
    out[4] = (int)fp_d2;
    out[5] = counts[0];  // |pred & gt & not_ignore|
    out[6] = counts[1];  // |(pred | gt) & not_ignore|
    out[7] = 0;
}

__global__ __launch_bounds__(256) void threshold_kernel(const float* __restrict__ probs, unsigned char* __restrict__ mask,
                                                        float thr, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) mask[i] = probs[i] > thr;
}

}  // namespace

extern "C" long isp_robot_click_workspace_bytes(int H, int W) {
    if (H <= 0 || W <= 0) return ISP_ERR_INVALID;
    return 2L * H * W * 4 + 64;  // g[2][H][W] int32 + keys[2] + counts[2]
}

extern "C" int isp_threshold_u8(const float* probs, void* mask, float thr, long n, void* stream) {
    ISP_CHECK_ARG(probs && mask && n > 0);
    threshold_kernel<<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(probs, (unsigned char*)mask, thr, n);
    return isp_launch_status();
}

extern "C" int isp_robot_click(const void* pred, const void* gt, const void* not_ignore, const void* not_clicked, int H,
                               int W, void* workspace, int* out, void* stream) {
    ISP_CHECK_ARG(pred && gt && not_clicked && workspace && out && H > 0 && W > 0 && W <= MAXW);
    ISP_CHECK_ARG((long)H * W < 0x7fffffffL && (long)H + W < 30000);  // squared distances stay in int32
    hipStream_t s = (hipStream_t)stream;
    int* g = (int*)workspace;
    unsigned long long* keys = (unsigned long long*)((char*)workspace + 2L * H * W * 4);
    int* counts = (int*)(keys + 2);
    if (hipMemsetAsync(keys, 0, 32, s) != hipSuccess) return ISP_ERR_LAUNCH;
    if (H <= 1024) {
        const int SEG = H <= 512 ? 32 : 64, nseg = (H + SEG - 1) / SEG;
        clicker_columns_seg_kernel<<<(W + 63) / 64, 64 * nseg, 0, s>>>((const unsigned char*)pred, (const unsigned char*)gt,
                                                                       (const unsigned char*)not_ignore, g, counts, H, W, SEG);
    } else {
        clicker_columns_kernel<<<(W + 63) / 64, 64, 0, s>>>((const unsigned char*)pred, (const unsigned char*)gt,
                                                            (const unsigned char*)not_ignore, g, counts, H, W);
    }
    clicker_rows_kernel<<<dim3(H, 2), 256, 0, s>>>(g, (const unsigned char*)not_clicked, keys, H, W);
    clicker_decide_kernel<<<1, 64, 0, s>>>(keys, counts, W, out);
    return isp_launch_status();
}
