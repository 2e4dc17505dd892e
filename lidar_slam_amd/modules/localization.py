"""Scan-matching driver (reference: modules/localization.py:95-134), batched.

The reference calls ``run_icp`` once per consecutive scan pair inside a Python loop.  All those
problems are independent — the initial guess comes from the odometry poses only
(localization.py:116-118) — so here the whole sequence is ONE ``lslam_icp_batch`` call; only the
float64 chain product of the relative poses stays sequential on the host (localization.py:127-130).
"""
from __future__ import annotations

import numpy as np

from ..batch import PackedScans, icp_batch, pack_clouds
from .utils import TSE2_from_TSE3, T_from_pose, pose_from_T  # noqa: F401  (re-exported like the reference)


def relative_pose_guesses(x_ts: np.ndarray) -> np.ndarray:
    """(N,3) odometry poses -> (N-1,4,4): inv(T(x[i-1])) @ T(x[i]) embedded in SE(3)
    (localization.py:116-118,121; vectorised closed form of the 3×3 inverse)."""
    x_ts = np.asarray(x_ts, dtype=np.float64)
    xa, ya, tha = x_ts[:-1].T
    xb, yb, thb = x_ts[1:].T
    ca, sa = np.cos(tha), np.sin(tha)
    dth = thb - tha
    dx, dy = xb - xa, yb - ya
    T = np.zeros((x_ts.shape[0] - 1, 4, 4))
    T[:, 0, 0] = np.cos(dth); T[:, 0, 1] = -np.sin(dth)
    T[:, 1, 0] = np.sin(dth); T[:, 1, 1] = np.cos(dth)
    T[:, 0, 3] = ca * dx + sa * dy
    T[:, 1, 3] = -sa * dx + ca * dy
    T[:, 2, 2] = 1.0
    T[:, 3, 3] = 1.0
    return T


def chain_poses(relative_T3: np.ndarray) -> np.ndarray:
    """Sequential float64 product of (S,3,3) relative transforms from the identity -> (S+1,3) poses
    (localization.py:111-130).  Runs in the library's host code (``lslam_chain_se2``): the chain is
    sequential by nature, 4955 tiny matmuls cost more in the Python interpreter than the whole GPU stage."""
    from .._lib import check, lib, ptr
    rel = np.ascontiguousarray(relative_T3, dtype=np.float64).reshape(-1, 3, 3)
    poses = np.empty((rel.shape[0] + 1, 3), dtype=np.float64)
    check(lib().lslam_chain_se2(ptr(rel), rel.shape[0], ptr(poses)))
    return poses


def scan_matching_relative_poses(x_ts, z_ts, device=0, return_info=False):
    """The batched part: (S,3,3) relative SE(2) transforms of scan i w.r.t. scan i-1."""
    x_ts = np.asarray(x_ts, dtype=np.float64)
    S = x_ts.shape[0] - 1
    if len(z_ts) != S + 1:
        raise ValueError("need one scan per pose")
    from ..scans import ScanSet
    if isinstance(z_ts, ScanSet):                      # points already in HBM: no packing, no upload
        r = z_ts.match_sequence(relative_pose_guesses(x_ts), epsilon=0.001)
        rel = np.tile(np.eye(3), (S, 1, 1))
        rel[:, :2, :2] = r.T[:, :2, :2]
        rel[:, :2, 2] = r.T[:, :2, 3]
        return (rel, r) if return_info else rel
    # a plain list of scans is packed straight into the library's pinned staging buffer by the C helper (a few threads,
    # finiteness checked on the way); an already packed list is used as it is
    pts, off = pack_clouds(z_ts if isinstance(z_ts, (PackedScans, list)) else [np.asarray(z) for z in z_ts], staging=0)
    if pts.shape[1] != 2:
        raise ValueError("scans must be (n_i, 2) robot-frame points")
    # source = scan i, target = scan i-1 (localization.py:120-121)
    r = icp_batch(pts, off[1:], pts, off[:-1], relative_pose_guesses(x_ts), epsilon=0.001, device=device)
    rel = np.tile(np.eye(3), (S, 1, 1))
    rel[:, :2, :2] = r.T[:, :2, :2]
    rel[:, :2, 2] = r.T[:, :2, 3]
    return (rel, r) if return_info else rel


def loop_closure_constraints(z_ts, n_poses=None, fixed_interval=10, max_distance=np.inf, max_yaw=np.inf,
                             device=0):
    """The loop-closure ICP problems of ``--mode gtsam`` (main.py:158-172) as ONE batched launch.

    For i = 0, K, 2K, … < n_poses − K: ``run_icp(z_ts[i], z_ts[i+K], return_error=True,
    normalize_error=True)`` — identity initial guess, default epsilon 0.01 — then the reference's
    gate ``translation < max_distance and rad2deg(angle) < max_yaw`` (main.py:167; the reference
    takes both limits from the odometry, main.py:100-101).

    Returns (pairs (L,2) int, T_se2 (L,3,3), errors (L,), accepted (L,) bool): every tested pair
    with its relative SE(2) transform and normalised error; ``accepted`` marks the ones the
    reference would add to the factor graph as BetweenFactorPose2(i, i+K, pose_from_T(T))."""
    n_poses = len(z_ts) if n_poses is None else int(n_poses)
    K = int(fixed_interval)
    starts = np.arange(0, n_poses - K, K, dtype=np.int64)
    if starts.size == 0:
        return np.zeros((0, 2), np.int64), np.zeros((0, 3, 3)), np.zeros(0), np.zeros(0, bool)
    src, so = pack_clouds([np.asarray(z_ts[i]) for i in starts], staging=0)
    tgt, to = pack_clouds([np.asarray(z_ts[i + K]) for i in starts], staging=1)
    r = icp_batch(src, so, tgt, to, np.tile(np.eye(4), (starts.size, 1, 1)), normalize_error=True, device=device)
    T3 = np.tile(np.eye(3), (starts.size, 1, 1))
    T3[:, :2, :2] = r.T[:, :2, :2]
    T3[:, :2, 2] = r.T[:, :2, 3]
    angle = np.arctan2(T3[:, 1, 0], T3[:, 0, 0])
    translation = np.linalg.norm(T3[:, :2, 2], axis=1)
    accepted = (translation < max_distance) & (np.rad2deg(angle) < max_yaw)
    return np.column_stack((starts, starts + K)), T3, r.error, accepted


def save_scan_matching_outputs(poses, relative_poses, dataset_num, out_dir="outputs"):
    """The two files ``--mode scan_matching`` leaves behind (main.py:123-124; consumed by
    plot_trajectories.py:13): ``poses_scan_matching_<ds>.npy`` (N,3) and
    ``relative_poses_scan_matching_<ds>.npy`` (N-1,3,3), float64."""
    import os
    os.makedirs(out_dir, exist_ok=True)
    paths = (os.path.join(out_dir, f"poses_scan_matching_{dataset_num}.npy"),
             os.path.join(out_dir, f"relative_poses_scan_matching_{dataset_num}.npy"))
    for path, arr in zip(paths, (poses, relative_poses)):
        with open(path, "wb") as f:
            np.save(f, np.asarray(arr, dtype=np.float64))
    return paths


# ---------------------------------------------------------------------------------------------
# LiDAR pre-processing (modules/localization.py:201-307), on the device
LIDAR_OFFSET = (0.13323, 0.0)             # p_rl[:2], localization.py:297
_ANGLE_MIN, _ANGLE_MAX = -135 * (np.pi / 180), 135 * (np.pi / 180)


def _split(xy, off):
    return [xy[off[k]:off[k + 1]] for k in range(off.shape[0] - 1)]


def _pack64(z_ts):
    off = np.zeros(len(z_ts) + 1, dtype=np.int64)
    arrs = [np.asarray(z, dtype=np.float64).reshape(-1, 2) for z in z_ts]
    off[1:] = np.cumsum([a.shape[0] for a in arrs])
    flat = np.ascontiguousarray(np.concatenate(arrs, axis=0)) if arrs else np.zeros((0, 2))
    return flat, off


def preprocess_lidar(lidar_ranges=None, lidar_range_min=0.1, lidar_range_max=30.0, z_ts=None, dbscan=None,
                     k_std=None, device=0, return_stats=False, resident=False):
    """The whole chain of main.py:84-87 in one library call, intermediate results staying on the GPU:
    ``get_lidar_data`` (when ``lidar_ranges`` (K, B) is given, else start from ``z_ts``) →
    ``DBSCAN_outliers_removal`` (``dbscan=(eps, min_samples)``) → ``statistical_outliers_removal``
    (``k_std``).  Returns the list of (n_k, 2) float64 scans [, (mean, std, threshold)]; with
    ``resident=True`` a :class:`lidar_slam_amd.scans.ScanSet` whose points never leave the device
    unless they are read."""
    if resident:
        from ..scans import ScanSet
        zs = (ScanSet.from_ranges(lidar_ranges, lidar_range_min, lidar_range_max, dbscan, k_std, device)
              if lidar_ranges is not None else ScanSet.from_scans(z_ts, dbscan, k_std, device))
        return (zs, zs.stats) if return_stats else zs
    from .. import _lib
    from .._lib import check, lib, ptr
    stages = 0
    ranges = cos_a = sin_a = in_xy = in_off = None
    K = B = 0
    if lidar_ranges is not None:
        ranges = np.ascontiguousarray(lidar_ranges, dtype=np.float64)
        K, B = ranges.shape
        angles = np.linspace(_ANGLE_MIN, _ANGLE_MAX, B)            # localization.py:272-278
        cos_a, sin_a = np.cos(angles), np.sin(angles)
        stages |= _lib.PRE_POLAR
        cap = K * B
    else:
        in_xy, in_off = _pack64(z_ts)
        K = in_off.shape[0] - 1
        cap = int(in_off[-1])
    eps, min_samples = (dbscan if dbscan is not None else (0.0, 0))
    if dbscan is not None:
        stages |= _lib.PRE_DBSCAN
    if k_std is not None:
        stages |= _lib.PRE_RADIAL
    out_xy = np.empty((max(cap, 1), 2), dtype=np.float64)
    out_off = np.zeros(K + 1, dtype=np.int64)
    stats = np.zeros(3, dtype=np.float64)
    check(lib().lslam_lidar_preprocess(stages, ptr(ranges), K, B, ptr(cos_a), ptr(sin_a), LIDAR_OFFSET[0], LIDAR_OFFSET[1],
                                       float(lidar_range_min), float(lidar_range_max), ptr(in_xy), ptr(in_off),
                                       float(eps), int(min_samples), float(k_std if k_std is not None else 0.0),
                                       ptr(out_xy), ptr(out_off), ptr(stats), int(device)))
    scans = PackedScans(out_xy[: int(out_off[-1])], out_off)     # list of views of one packed array
    return (scans, tuple(stats)) if return_stats else scans


def get_lidar_data(lidar_ranges, lidar_range_min, lidar_range_max):
    """(N, 1081) ranges → list of (n_i, 2) robot-frame points: beams over [−135°, 135°], ranges
    outside [min, max] dropped, lidar→body offset added (modules/localization.py:252-307)."""
    return preprocess_lidar(lidar_ranges, lidar_range_min, lidar_range_max)


def DBSCAN_outliers_removal(z_ts, eps=0.5, min_samples=10):
    """Drop the points sklearn's ``DBSCAN(eps, min_samples)`` labels −1
    (modules/localization.py:201-221): a point stays iff it is a core point or within eps of one."""
    print(f"DBSCAN: eps={eps}, min_samples={min_samples}")
    return preprocess_lidar(z_ts=z_ts, dbscan=(eps, min_samples))


def statistical_outliers_removal(z_ts, k_std=3):
    """Drop the points farther than mean + k_std·std of all ranges (modules/localization.py:223-250)."""
    print(f"Statistical outliers removal: k_std={k_std}")
    scans, (mean, std, thr) = preprocess_lidar(z_ts=z_ts, k_std=k_std, return_stats=True)
    print(f"Mean distance: {mean:.3f}")
    print(f"Std distance: {std:.3f}")
    print(f"Threshold distance: {thr:.3f}")
    return scans


def poses_from_scan_matching(x_ts, z_ts, return_relative_poses=False):
    """Refine the odometry trajectory by frame-to-frame scan matching
    (modules/localization.py:95-134): same arguments, same returns —
    ``poses (N,3)`` [, ``relative_poses (N-1,3,3)``]."""
    rel = scan_matching_relative_poses(x_ts, z_ts)
    poses = chain_poses(rel)
    if return_relative_poses:
        return poses, rel
    return poses
