/* Oracle (test infrastructure only): C restatement of the reference's single
 * native component, get_dist_maps (core/utils/cython/_get_dist_maps.pyx:18-64).
 *
 * Multi-source FIFO flood over the 4-connected grid.  Every queue entry carries
 * the seed it descends from; a neighbour is (re)queued when the squared,
 * normalised distance to that seed improves on the stored value.  Layer 0 =
 * first half of the point rows (positive clicks), layer 1 = second half
 * (pyx:38-41).  Seeds are the clicks rounded with Python's round(), i.e.
 * half-to-even (pyx:31 calls the builtin on a Python float) -> nearbyint under
 * the default FE_TONEAREST mode.  A row with rounded row < 0 is skipped (pyx:32).
 *
 * Unlike the reference (no bounds checks, pyx:15-17,42) a seed outside the grid
 * returns -2 instead of writing out of bounds.
 */
#include <math.h>
#include <stdlib.h>

typedef struct { int row, col, layer, orig_row, orig_col; } qnode;

int oracle_get_dist_maps(const float* points, int npoints, int height, int width,
                         float norm_delimeter, float* dist_maps /* [2,H,W] */) {
    const long plane = (long)height * width;
    for (long i = 0; i < 2 * plane; ++i) dist_maps[i] = 1e6f;      /* pyx:22-23 */
    qnode* q = (qnode*)malloc((4 * plane + 1) * sizeof(qnode));    /* pyx:26 */
    if (!q) return -1;
    long head = 0, tail = -1;
    static const int dxy[8] = {-1, 0, 0, -1, 0, 1, 1, 0};           /* pyx:25 */

    for (int i = 0; i < npoints; ++i) {                              /* pyx:30-42 */
        int x = (int)nearbyint((double)points[3 * i + 0]);
        int y = (int)nearbyint((double)points[3 * i + 1]);
        if (x < 0) continue;
        if (x >= height || y < 0 || y >= width) { free(q); return -2; }
        qnode n;
        n.row = n.orig_row = x;
        n.col = n.orig_col = y;
        n.layer = (2 * i >= npoints) ? 1 : 0;      /* i >= shape[0] / 2 (true division) */
        q[++tail] = n;
        dist_maps[n.layer * plane + (long)x * width + y] = 0.f;
    }

    while (tail - head + 1 > 0) {                                    /* pyx:44-61 */
        qnode v = q[head++];
        for (int k = 0; k < 4; ++k) {
            int x = v.row + dxy[2 * k], y = v.col + dxy[2 * k + 1];
            /* Cython evaluates (int - int) / float in C double?  No: operands are
             * C int and C float -> float division, then ** 2 -> float multiply. */
            float a = (float)(x - v.orig_row) / norm_delimeter;
            float b = (float)(y - v.orig_col) / norm_delimeter;
            float ndist = a * a + b * b;
            if (x >= 0 && y >= 0 && x < height && y < width) {
                float* cell = &dist_maps[v.layer * plane + (long)x * width + y];
                if (*cell > ndist) {
                    if (tail + 1 >= 4 * plane + 1) { free(q); return -3; }
                    qnode n = v;
                    n.row = x; n.col = y;
                    q[++tail] = n;
                    *cell = ndist;
                }
            }
        }
    }
    free(q);
    return 0;
}
