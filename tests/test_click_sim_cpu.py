"""CPU: the oracle's restatement of cv2.distanceTransform(mask, DIST_L2, 5) (oracle/chamfer5.c) against
known answers of OpenCV's 5x5 chamfer mask, and the oracle's get_next_points selection logic."""
import numpy as np
from scipy.ndimage import distance_transform_edt

from oracle import click_simulation as osim


def test_chamfer5_known_answers():
    m = np.ones((7, 7), np.uint8)
    m[3, 3] = 0
    d = osim.chamfer5(m)
    a, b, c = 1.0, 1.4, 2.1969  # DIST_L2 / 5x5 step costs (axial, diagonal, knight move)
    fix = lambda *steps: np.float32(sum(int(round(np.float32(s) * 65536)) for s in steps) / 65536.0)
    assert d[3, 3] == 0 and d[3, 4] == fix(a) and d[2, 4] == fix(b) and d[1, 4] == fix(c)
    assert d[1, 5] == fix(b, b) and d[0, 4] == fix(c, a) and d[0, 5] == fix(c, b) and d[0, 6] == fix(b, b, b)
    assert np.array_equal(d, d.T) and np.array_equal(d, d[::-1]) and np.array_equal(d, d[:, ::-1])


def test_chamfer5_close_to_exact_edt():
    rng = np.random.default_rng(0)
    yy, xx = np.mgrid[:90, :120]
    m = np.zeros((90, 120), bool)
    for _ in range(4):
        cy, cx, ry, rx = rng.integers(0, 90), rng.integers(0, 120), rng.integers(5, 40), rng.integers(5, 50)
        m |= ((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 <= 1
    m = np.pad(m, 1)
    d, e = osim.chamfer5(m), distance_transform_edt(m)
    assert np.all((d == 0) == (e == 0))
    assert np.abs(d - e).max() <= 0.03 * e.max() + 1e-3  # the 5x5 chamfer metric is within ~2 % of Euclidean
    assert osim.chamfer5(np.zeros((5, 6), np.uint8)).max() == 0


def test_get_next_points_selection():
    gt = np.zeros((1, 1, 20, 30), np.float32)
    gt[0, 0, 5:15, 8:22] = 1
    pred = np.zeros_like(gt)            # everything missed -> positive click inside the object
    pts = -np.ones((1, 6, 3), np.float32)
    out = osim.get_next_points(pred, gt, pts, 1, np.array([0]))
    assert out[0, 2, 2] == 1 and 5 <= out[0, 2, 0] < 15 and 8 <= out[0, 2, 1] < 22  # slot P - click_indx
    assert np.all(out[0, [0, 1, 3, 4, 5]] == -1)
    first = out[0, 2, :2].copy()
    last = osim.get_next_points(pred, gt, pts, 1, np.array([2 ** 32 - 1]))[0, 2, :2]
    assert tuple(first) < tuple(last)   # row-major order of the inner set
    pred2 = np.ones_like(gt)            # everything predicted -> negative click outside the object
    out2 = osim.get_next_points(pred2, gt, pts, 2, np.array([123456789]))
    assert out2[0, 4, 2] == 2 and gt[0, 0, int(out2[0, 4, 0]), int(out2[0, 4, 1])] == 0  # slot 2P - click_indx
    same = osim.get_next_points(gt.copy(), gt, pts, 1, np.array([5]))  # perfect prediction: unchanged
    assert np.array_equal(same, pts)


def test_chamfer5_is_the_exact_graph_distance_of_the_5x5_mask():
    """An independent derivation of what the two raster passes compute: the shortest-path distance to the nearest zero pixel on
    the grid graph whose edges are the 5x5 chamfer mask's 16 moves (4 axial at a, 4 diagonal at b, 8 knight moves at c, integer
    costs in 2^-16 units), paths confined to the image.  Dijkstra from all zero pixels at once; the restatement must equal it
    exactly -- random masks with holes, blobs, border contact and single-pixel features."""
    import heapq
    a, b, c = (int(np.float32(v) * np.float32(65536.0) + np.float32(0.5)) for v in (1.0, 1.4, 2.1969))
    moves = [(0, 1, a), (0, -1, a), (1, 0, a), (-1, 0, a), (1, 1, b), (1, -1, b), (-1, 1, b), (-1, -1, b)]
    moves += [(dy, dx, c) for dy, dx in ((1, 2), (1, -2), (-1, 2), (-1, -2), (2, 1), (2, -1), (-2, 1), (-2, -1))]

    def dijkstra(mask):
        h, w = mask.shape
        dist = np.full((h, w), np.iinfo(np.int64).max, np.int64)
        heap = []
        for y, x in zip(*np.nonzero(mask == 0)):
            dist[y, x] = 0
            heap.append((0, int(y), int(x)))
        heapq.heapify(heap)
        while heap:
            d, y, x = heapq.heappop(heap)
            if d > dist[y, x]:
                continue
            for dy, dx, cost in moves:
                yy, xx = y + dy, x + dx
                if 0 <= yy < h and 0 <= xx < w and d + cost < dist[yy, xx]:
                    dist[yy, xx] = d + cost
                    heapq.heappush(heap, (d + cost, yy, xx))
        return dist

    rng = np.random.default_rng(7)
    for trial in range(6):
        h, w = int(rng.integers(9, 40)), int(rng.integers(9, 48))
        yy, xx = np.mgrid[:h, :w]
        m = np.zeros((h, w), bool)
        for _ in range(int(rng.integers(1, 5))):
            cy, cx, ry, rx = rng.integers(0, h), rng.integers(0, w), rng.integers(2, 14), rng.integers(2, 16)
            m |= ((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 <= 1
        m &= rng.random((h, w)) > 0.02 * trial          # pinholes inside the blobs
        if trial % 2 == 0:
            m = np.pad(m, 1)[: h, : w] | m               # (some masks touch the border, some do not)
        if m.all():
            m[h // 2, w // 2] = False
        got = osim.chamfer5(m.astype(np.uint8))
        want = dijkstra(m.astype(np.uint8))
        assert np.array_equal(got, (want.astype(np.float64) / 65536.0).astype(np.float32)), trial
