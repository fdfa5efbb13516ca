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
