"""Scan-matching driver (reference: modules/localization.py:95-134), batched.

The reference calls ``run_icp`` once per consecutive scan pair inside a Python loop.  All those
problems are independent — the initial guess comes from the odometry poses only
(localization.py:116-118) — so here the whole sequence is ONE ``lslam_icp_batch`` call; only the
float64 chain product of the relative poses stays sequential on the host (localization.py:127-130).
"""
from __future__ import annotations

import numpy as np

from ..batch import icp_batch, pack_clouds
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
    (localization.py:111-130)."""
    T = np.eye(3)
    poses = np.zeros((relative_T3.shape[0] + 1, 3))
    for i, Trel in enumerate(relative_T3):
        T = T @ Trel
        poses[i + 1] = (T[0, 2], T[1, 2], np.arctan2(T[1, 0], T[0, 0]))
    return poses


def scan_matching_relative_poses(x_ts, z_ts, device=0, return_info=False):
    """The batched part: (S,3,3) relative SE(2) transforms of scan i w.r.t. scan i-1."""
    x_ts = np.asarray(x_ts, dtype=np.float64)
    S = x_ts.shape[0] - 1
    if len(z_ts) != S + 1:
        raise ValueError("need one scan per pose")
    pts, off = pack_clouds([np.asarray(z) for z in z_ts])
    if pts.shape[1] != 2:
        raise ValueError("scans must be (n_i, 2) robot-frame points")
    # source = scan i, target = scan i-1 (localization.py:120-121)
    r = icp_batch(pts, off[1:], pts, off[:-1], relative_pose_guesses(x_ts), epsilon=0.001, device=device)
    rel = np.tile(np.eye(3), (S, 1, 1))
    rel[:, :2, :2] = r.T[:, :2, :2]
    rel[:, :2, 2] = r.T[:, :2, 3]
    return (rel, r) if return_info else rel


def poses_from_scan_matching(x_ts, z_ts, return_relative_poses=False):
    """Refine the odometry trajectory by frame-to-frame scan matching
    (modules/localization.py:95-134): same arguments, same returns —
    ``poses (N,3)`` [, ``relative_poses (N-1,3,3)``]."""
    rel = scan_matching_relative_poses(x_ts, z_ts)
    poses = chain_poses(rel)
    if return_relative_poses:
        return poses, rel
    return poses
