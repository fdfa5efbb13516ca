"""Oracle: click -> disk / distance maps (test infrastructure only).

Follows the torch path of ``DistMaps.get_coord_features``
(reference core/model/ops.py:35-77) op for op in IEEE fp32, but keeps a running
minimum over the clicks instead of materialising the [B*2P, 2, H, W] temporary
(min is exact, so the result is bit-identical).
"""
import ctypes
import os

import numpy as np

_F = np.float32


def click_maps(points, rows, cols, norm_radius, spatial_scale=1.0, use_disks=False):
    """points: [B, 2P, 3] float32 (row, col, order); first P rows positive,
    last P negative; a row whose max(row, col) < 0 is empty (ops.py:40).
    Returns [B, 2, rows, cols] float32."""
    pts = np.asarray(points, dtype=_F)
    B, P2, _ = pts.shape
    P = P2 // 2
    scale = _F(spatial_scale)
    rr = np.arange(rows, dtype=_F)[:, None]
    cc = np.arange(cols, dtype=_F)[None, :]
    # ops.py:60 divides by the *python* double norm_radius*spatial_scale; for an fp32
    # tensor torch casts that scalar to fp32 first.
    denom = _F(float(norm_radius) * float(spatial_scale))
    out = np.empty((B, 2, rows, cols), dtype=_F)
    for b in range(B):
        for pol in range(2):
            best = np.full((rows, cols), _F(1e6), dtype=_F)
            for k in range(P):
                pr, pc = pts[b, pol * P + k, 0], pts[b, pol * P + k, 1]
                if max(pr, pc) < 0:  # invalid row -> 1e6 (ops.py:40,66)
                    continue
                dr = rr + (-(pr * scale))  # coords.add_(-add_xy)  ops.py:55-58
                dc = cc + (-(pc * scale))
                if not use_disks:
                    dr = dr / denom  # ops.py:59-60
                    dc = dc / denom
                d = dr * dr + dc * dc  # ops.py:61-63
                best = np.minimum(best, d.astype(_F))  # ops.py:68-69
            out[b, pol] = best
    if use_disks:
        thr = _F((float(norm_radius) * float(spatial_scale)) ** 2)  # ops.py:72-73
        return (out <= thr).astype(_F)
    # ops.py:75  sqrt_().mul_(2).tanh_()
    return np.tanh(np.sqrt(out) * _F(2)).astype(_F)


# ---------------------------------------------------------------------------
# BFS restatement of the reference's only native component
# (core/utils/cython/_get_dist_maps.pyx:18-64), compiled from oracle/dist_maps_bfs.c
# ---------------------------------------------------------------------------
_LIB = None


def _bfs_lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libdist_maps_bfs.so")
        if not os.path.exists(path):
            raise RuntimeError(
                f"{path} not built; run `make -C oracle` or __graft_entry__.build()"
            )
        lib = ctypes.CDLL(path)
        lib.oracle_get_dist_maps.restype = ctypes.c_int
        lib.oracle_get_dist_maps.argtypes = [
            ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
            ctypes.c_float, ctypes.c_void_p,
        ]
        _LIB = lib
    return _LIB


def get_dist_maps_bfs(points, height, width, norm_delimeter):
    """points: [2P, 3] float32 C-contiguous -> [2, H, W] float32 squared distances
    (1e6 where unreachable), exactly as _get_dist_maps.pyx:18-64."""
    pts = np.ascontiguousarray(points, dtype=_F)
    out = np.empty((2, height, width), dtype=_F)
    rc = _bfs_lib().oracle_get_dist_maps(
        pts.ctypes.data, pts.shape[0], height, width, float(norm_delimeter), out.ctypes.data
    )
    if rc != 0:
        raise RuntimeError(f"oracle_get_dist_maps failed rc={rc}")
    return out


def click_maps_cpu_mode(points, rows, cols, norm_radius, spatial_scale=1.0, use_disks=False):
    """The ``cpu_mode=True`` branch of DistMaps (ops.py:21-34 + :72-75): per-sample
    BFS on rounded clicks, then the same disk / tanh post-processing."""
    pts = np.asarray(points, dtype=_F)
    delim = 1.0 if use_disks else float(spatial_scale) * float(norm_radius)  # ops.py:24-26
    maps = np.stack([get_dist_maps_bfs(pts[b], rows, cols, delim) for b in range(pts.shape[0])])
    if use_disks:
        thr = _F((float(norm_radius) * float(spatial_scale)) ** 2)
        return (maps <= thr).astype(_F)
    return np.tanh(np.sqrt(maps) * _F(2)).astype(_F)
