"""Point-to-point ICP on MI355X behind the call surface of the reference's ``modules/icp.py``.

Every function keeps the reference's name, argument order, defaults and return convention
(modules/icp.py:4,29,43,76,100,123-133); the arithmetic runs in the HIP kernels of
``liblslam_hip.so`` (no CPU fallback — without the library or a GPU the calls raise).

Differences a caller can observe: none beyond float64 summation order.  The device searches in
float32 but decides every near tie on the float64 points (float64-exact mode), so the correspondences are those of
the reference's float64 search — lowest index on exact distance ties, where the KD-tree picks an arbitrary one —
and moments, solve, error and stop rules are float64 on the float64 points: T and error agree with the reference to
~1e-12.  (float32 input arrays are searched as they are: exact float32 nearest neighbours.)
"""
from __future__ import annotations

import numpy as np

from .. import _lib
from .._lib import check, lib, ptr
from ..batch import icp_batch

# one workgroup iterates a pair on chip; beyond this many pair evaluations per iteration the
# whole-chip path (grid-wide LDS-tiled search; one launch ≈ 12 µs per iteration below 8192 source
# points, three launches above) is faster
LARGE_PAIR_EVALS = 300_000

_DEVICE = 0

# Search strategy of the nearest-neighbour step (LSLAM_ICP_PRUNE / LSLAM_ICP_ALL_PAIRS in include/lslam.h): None
# lets the library pick, True forces the exact spatial pruning, False the all-pairs sweep.  Same correspondences
# and iteration counts either way; the reference's signatures stay untouched, so it is a module switch rather
# than an argument.
PRUNE = None


def set_device(device: int) -> None:
    """GPU used by the calls of this module (one process per GPU is the intended deployment)."""
    global _DEVICE
    check(lib().lslam_set_device(int(device)))
    _DEVICE = int(device)


class TargetCloud:
    """Stand-in for the reference's ``KDTree(pc2[:, :3])`` (modules/icp.py:161): just the target
    points — brute force needs no index structure.  ``.data`` mirrors ``KDTree.data``."""

    def __init__(self, points):
        pts = np.asarray(points, dtype=np.float64)
        if pts.ndim != 2 or pts.shape[1] < 2:
            raise ValueError("target points must be (M,2) or (M,3+)")
        self.data = _xyz(pts)


def _xyz(pc):
    """first three columns, z = 0 appended to planar input (modules/icp.py:153-156)"""
    pc = np.asarray(pc)
    if pc.ndim != 2 or pc.shape[1] < 2:
        raise ValueError("point cloud must be (N,2) or (N,3)")
    if pc.shape[1] == 2:
        return np.hstack((pc, np.zeros((pc.shape[0], 1))))
    return pc[:, :3]


def _keeps_planar(T):
    return T[2, 0] == 0.0 and T[2, 1] == 0.0 and T[2, 3] == 0.0


def voxel_downsample(point_cloud, voxel_size):
    """Mean of the points inside every occupied voxel, voxels in lexicographic order of their integer
    index ``floor((p - min) / voxel_size)`` (modules/icp.py:4-27).  On the device: float64 keys, stable
    radix sort, one thread per voxel (the reference loops over voxels in Python, O(V·N))."""
    pc = np.ascontiguousarray(point_cloud, dtype=np.float64)
    if pc.ndim != 2 or pc.shape[1] not in (2, 3):
        raise ValueError("point cloud must be (N,2) or (N,3)")
    out = np.empty_like(pc)
    count = np.zeros(1, dtype=np.int64)
    check(lib().lslam_voxel_downsample(ptr(pc), pc.shape[0], pc.shape[1], float(voxel_size), ptr(out), ptr(count), _DEVICE))
    return out[: int(count[0])].copy()


def get_correspondences(pc1, pc2_tree):
    """Index of the nearest target point for every row of pc1 (modules/icp.py:29-41).
    ``pc2_tree`` is a :class:`TargetCloud` (or anything with a ``.data`` array of points, such as
    the ``scipy.spatial.KDTree`` the reference builds)."""
    data, pc1 = np.asarray(pc2_tree.data), np.asarray(pc1)
    out = np.empty(pc1.shape[0], dtype=np.int32)
    if data.dtype == np.float32 and pc1.dtype == np.float32:      # float32 points: searched as they are
        tgt = np.ascontiguousarray(data[:, :3], dtype=np.float32)
        src = np.ascontiguousarray(_xyz(pc1), dtype=np.float32)
        check(lib().lslam_nn(ptr(src), src.shape[0], ptr(tgt), tgt.shape[0], 3, None, ptr(out), None, _DEVICE))
    else:                                                         # the reference's type: float64-exact mode
        tgt = np.ascontiguousarray(data[:, :3], dtype=np.float64)
        src = np.ascontiguousarray(_xyz(pc1), dtype=np.float64)
        check(lib().lslam_nn_f64(ptr(src), src.shape[0], ptr(tgt), tgt.shape[0], 3, None, ptr(out), _DEVICE))
    return out.astype(np.int64)


def get_transform(pc1, pc2):
    """Kabsch rigid transform taking the rows of pc1 onto the rows of pc2 (modules/icp.py:43-74)."""
    p = np.ascontiguousarray(np.asarray(pc1, dtype=np.float64)[:, :3])
    q = np.ascontiguousarray(np.asarray(pc2, dtype=np.float64)[:, :3])
    if p.shape != q.shape:
        raise ValueError("pc1 and pc2 must pair row by row")
    T = np.empty((4, 4), dtype=np.float64)
    check(lib().lslam_kabsch(ptr(p), 3, ptr(q), 3, p.shape[0], ptr(T), _DEVICE))
    return T


def get_error(pc1, pc2, normalize=False):
    """Σ‖pc1 − pc2‖² over paired rows; with ``normalize`` divided by (joint bounding-box
    diagonal)² · N (modules/icp.py:76-98)."""
    p = np.ascontiguousarray(pc1, dtype=np.float64)
    q = np.ascontiguousarray(pc2, dtype=np.float64)
    if p.shape != q.shape or p.ndim != 2 or not 1 <= p.shape[1] <= 3:
        raise ValueError("pc1 and pc2 must both be (N,d) with d <= 3")
    out = np.empty(1, dtype=np.float64)
    check(lib().lslam_sq_error(ptr(p), ptr(q), p.shape[0], p.shape[1], int(bool(normalize)), ptr(out), _DEVICE))
    return np.float64(out[0])


# float32 points resolve 6e-5 m at |coordinate| = 1000 m; clouds further out than this are shifted towards
# the origin in float64 before they are rounded (sensor-frame LiDAR scans never are: ±30 m)
CENTRE_ABOVE = 1000.0


def _solve(pc1, pc2, init_transform, epsilon, max_iters, stopping_thresh, normalize_error):
    """-> (T (4,4) f64, error f64, correspondences int64)"""
    src = np.asarray(pc1)
    tgt = np.asarray(pc2)
    if src.ndim != 2 or tgt.ndim != 2 or src.shape[1] < 2 or tgt.shape[1] < 2:
        raise ValueError("point clouds must be (N,2) or (N,3)")
    T0 = np.array(init_transform, dtype=np.float64).reshape(4, 4)
    if tgt.shape[0] and src.shape[0] and np.isfinite(T0).all():
        reach = max(np.abs(tgt[:, :3]).max(), np.abs(src[:, :3]).max())
        if np.isfinite(reach) and reach > CENTRE_ABOVE:
            # q ≈ R p + t  ⇔  q − c ≈ R (p − c) + (R c + t − c): solve in coordinates relative to the target's centroid
            d = min(src.shape[1], tgt.shape[1], 3)
            c = np.zeros(3)
            c[:d] = np.mean(np.asarray(tgt[:, :d], dtype=np.float64), axis=0)
            shift = lambda pc: np.asarray(pc[:, :3], dtype=np.float64) - c[: min(pc.shape[1], 3)]
            Tc = T0.copy()
            Tc[:3, 3] = T0[:3, :3] @ c + T0[:3, 3] - c
            T, err, corr = _solve_near(shift(src), shift(tgt), Tc, epsilon, max_iters, stopping_thresh, normalize_error)
            T[:3, 3] = T[:3, 3] + c - T[:3, :3] @ c
            return T, err, corr
    return _solve_near(src, tgt, T0, epsilon, max_iters, stopping_thresh, normalize_error)


def _solve_near(src, tgt, T0, epsilon, max_iters, stopping_thresh, normalize_error):
    """float64 clouds go to the float64-exact entry points (``lslam_icp_*_f64``): the device searches in float32 and
    decides every near tie on the float64 points, so the correspondences are those of the reference's float64 search
    (modules/icp.py:29-41).  Clouds that ARE float32 are searched as they are (exact float32 nearest neighbours)."""
    planar = src.shape[1] == 2 and tgt.shape[1] == 2 and _keeps_planar(T0)
    dtype = np.float32 if (src.dtype == np.float32 and tgt.dtype == np.float32) else np.float64
    if planar:
        s = np.ascontiguousarray(src, dtype=dtype)
        t = np.ascontiguousarray(tgt, dtype=dtype)
    else:
        s = np.ascontiguousarray(_xyz(src), dtype=dtype)
        t = np.ascontiguousarray(_xyz(tgt), dtype=dtype)
    n, m, dim = s.shape[0], t.shape[0], s.shape[1]
    if m == 0:
        raise IndexError("index 0 is out of bounds for axis 0 with size 0 (empty target cloud)")
    if n > 0 and n * m > LARGE_PAIR_EVALS:
        T = np.empty((4, 4), dtype=np.float64)
        err = np.empty(1, dtype=np.float64)
        it = np.empty(1, dtype=np.int32)
        corr = np.empty(n, dtype=np.int32)
        flags = (_lib.ICP_NORMALIZE_ERROR if normalize_error else 0) | _lib.search_flags(PRUNE)
        entry = lib().lslam_icp_large_f64 if dtype == np.float64 else lib().lslam_icp_large
        check(entry(ptr(s), n, ptr(t), m, dim, ptr(T0), float(epsilon), float(stopping_thresh), int(max_iters), flags,
                    ptr(T), ptr(err), ptr(it), ptr(corr), _DEVICE))
        return T, np.float64(err[0]), corr.astype(np.int64)
    r = icp_batch(s, np.array([0, n], dtype=np.int64), t, np.array([0, m], dtype=np.int64), T0[None],
                  epsilon=epsilon, max_iters=max_iters, stopping_thresh=stopping_thresh,
                  normalize_error=normalize_error, return_correspondences=True, device=_DEVICE, prune=PRUNE)
    return r.T[0], np.float64(r.error[0]), r.correspondences.astype(np.int64)


def icp_iteration(pc1, pc2, pc2_tree, previous_transform, normalize_error=False):
    """One ICP iteration (modules/icp.py:100-121): -> (T_step @ previous_transform,
    correspondences, error of the pre-update residual).  ``pc2_tree`` is accepted for signature
    parity; the search is brute force on pc2 itself."""
    T, err, corr = _solve(_xyz(pc1), _xyz(pc2), previous_transform, epsilon=0.0, max_iters=0,
                          stopping_thresh=0.0, normalize_error=normalize_error)
    return T, corr, err


def run_icp(
    pc1,
    pc2,
    init_transform=np.eye(4),
    epsilon=0.01,
    max_iters=2000,
    stopping_thresh=0.0001,
    return_error=False,
    normalize_error=False,
    return_correspondences=False,
):
    """Align pc1 (N,2|3) to pc2 (M,2|3) (modules/icp.py:123-189).

    Stop rules, in order, after every iteration: ``error < epsilon``; ``iteration >= max_iters``;
    ``|last_error - error| < stopping_thresh``.  Returns ``T`` | ``(T, error)`` |
    ``(T, correspondences)`` | ``(T, error, correspondences)`` by flag combination, exactly as the
    reference; correspondences are those of the last iteration, int64."""
    T, err, corr = _solve(pc1, pc2, init_transform, epsilon, max_iters, stopping_thresh, normalize_error)
    if return_error and not return_correspondences:
        return T, err
    if return_correspondences and not return_error:
        return T, corr
    if return_error and return_correspondences:
        return T, err, corr
    return T
