/* Oracle (test infrastructure only): cv2.distanceTransform(mask, cv2.DIST_L2, 5) restated in C.
 *
 * Call site in the reference: core/training/trainer.py:597-598 (train-time click simulation,
 * get_next_points), on the 1-pixel zero-padded FN / FP masks.  The arithmetic lives in OpenCV
 * (requirements.txt:  opencv-python==4.4.0.46), which is absent from the container, so this is a
 * restatement of OpenCV's published two-pass 5x5 chamfer transform (modules/imgproc/src/distransform.cpp,
 * distanceTransform_5x5): fixed point with DIST_SHIFT = 16, step costs a = 1, b = 1.4, c = 2.1969 (the
 * DIST_L2 / 5x5 metrics of getDistanceTransformMask) rounded to integers, a forward raster pass over the
 * causal half of the 5x5 neighbourhood, a backward pass over the other half, INT_MAX>>2 outside the image,
 * result (float)(t * 2^-16).  No reference test pins it: PARITY UNPINNED (see oracle/__init__.py).  What CAN be checked here is:
 * known answers of the mask, and that the two passes equal the exact shortest-path distance of the 5x5 mask's 16 moves
 * (Dijkstra, tests/test_click_sim_cpu.py) -- i.e. only the constants and the fixed-point format rest on the recollection. */
#include <limits.h>
#include <stdint.h>
#include <stdlib.h>

#define DIST_SHIFT 16
#define INIT_DIST0 ((uint32_t)(INT_MAX >> 2))

static uint32_t fix(float x) { return (uint32_t)(long)(x * (float)(1 << DIST_SHIFT) + 0.5f); }

/* mask: uint8 [h][w] (non-zero = inside), dist: float [h][w].  Returns 0, or -1 on allocation failure. */
int oracle_chamfer5(const uint8_t* mask, int h, int w, float* dist) {
    const int B = 2;
    const long step = w + 2 * B;
    const uint32_t HV = fix(1.0f), DIAG = fix(1.4f), LONG = fix(2.1969f);
    const float scale = 1.0f / (float)(1 << DIST_SHIFT);
    uint32_t* temp = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)(h + 2 * B) * step);
    if (!temp) return -1;
    for (long i = 0; i < (long)(h + 2 * B) * step; ++i) temp[i] = INIT_DIST0;
    for (int i = 0; i < h; ++i) { /* forward pass */
        const uint8_t* s = mask + (long)i * w;
        uint32_t* tmp = temp + (long)(i + B) * step + B;
        for (int j = 0; j < w; ++j) {
            if (!s[j]) {
                tmp[j] = 0;
            } else {
                uint32_t t0 = tmp[j - step * 2 - 1] + LONG, t;
                t = tmp[j - step * 2 + 1] + LONG; if (t0 > t) t0 = t;
                t = tmp[j - step - 2] + LONG;     if (t0 > t) t0 = t;
                t = tmp[j - step - 1] + DIAG;     if (t0 > t) t0 = t;
                t = tmp[j - step] + HV;           if (t0 > t) t0 = t;
                t = tmp[j - step + 1] + DIAG;     if (t0 > t) t0 = t;
                t = tmp[j - step + 2] + LONG;     if (t0 > t) t0 = t;
                t = tmp[j - 1] + HV;              if (t0 > t) t0 = t;
                tmp[j] = t0;
            }
        }
    }
    for (int i = h - 1; i >= 0; --i) { /* backward pass */
        float* d = dist + (long)i * w;
        uint32_t* tmp = temp + (long)(i + B) * step + B;
        for (int j = w - 1; j >= 0; --j) {
            uint32_t t0 = tmp[j];
            if (t0 > HV) {
                uint32_t t = tmp[j + step * 2 + 1] + LONG; if (t0 > t) t0 = t;
                t = tmp[j + step * 2 - 1] + LONG; if (t0 > t) t0 = t;
                t = tmp[j + step + 2] + LONG;     if (t0 > t) t0 = t;
                t = tmp[j + step + 1] + DIAG;     if (t0 > t) t0 = t;
                t = tmp[j + step] + HV;           if (t0 > t) t0 = t;
                t = tmp[j + step - 1] + DIAG;     if (t0 > t) t0 = t;
                t = tmp[j + step - 2] + LONG;     if (t0 > t) t0 = t;
                t = tmp[j + 1] + HV;              if (t0 > t) t0 = t;
                tmp[j] = t0;
            }
            if (t0 > INIT_DIST0) t0 = INIT_DIST0;
            d[j] = (float)t0 * scale;
        }
    }
    free(temp);
    return 0;
}
