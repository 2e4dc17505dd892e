"""SE(2)/SE(3) glue and the scan-matching driver, restated (test oracle).

TEST INFRASTRUCTURE — see ``oracle/__init__.py``.  Parity: PINNED by the
reference's committed results ``outputs/poses_odom_20.npy`` →
``outputs/relative_poses_odom_20.npy`` (get_relative_pose, 7e-15) and
``outputs/relative_poses_scan_matching_20.npy`` → ``poses_scan_matching_20.npy``
(chain product + pose_from_T, 4e-16); a 512-pose slice of each is committed as
``tests/golden/se2_outputs20.npz``.  The scan-matching driver is pinned by
``tests/golden/seq_scanmatch.npz``.
"""
from __future__ import annotations

import numpy as np

from . import icp_oracle


def T_from_pose(pose) -> np.ndarray:
    """(x, y, θ) -> 3×3 homogeneous SE(2) matrix (modules/utils.py:178-194)."""
    x, y, th = pose
    c, s = np.cos(th), np.sin(th)
    return np.array([[c, -s, x], [s, c, y], [0.0, 0.0, 1.0]])


def TSE2_from_TSE3(T4: np.ndarray) -> np.ndarray:
    """Keep the xy rotation block and xy translation (modules/utils.py:212-225)."""
    T3 = np.eye(3)
    T3[:2, :2] = T4[:2, :2]
    T3[:2, 2] = T4[:2, 3]
    return T3


def TSE3_from_TSE2(T3: np.ndarray) -> np.ndarray:
    """Embed SE(2) in a 4×4 identity (modules/utils.py:227-240)."""
    T4 = np.eye(4)
    T4[:2, :2] = T3[:2, :2]
    T4[:2, 3] = T3[:2, 2]
    return T4


def pose_from_T(T: np.ndarray) -> np.ndarray:
    """(x, y, atan2(T10, T00)) (modules/utils.py:196-210)."""
    if T.shape[0] == 4:
        T = TSE2_from_TSE3(T)
    return np.array([T[0, 2], T[1, 2], np.arctan2(T[1, 0], T[0, 0])])


def get_relative_pose(pose_a, pose_b) -> np.ndarray:
    """inv(T(pose_a)) @ T(pose_b): frame b expressed in frame a
    (modules/utils.py:161-176, duplicated at modules/localization.py:162-177)."""
    return np.dot(np.linalg.inv(T_from_pose(pose_a)), T_from_pose(pose_b))


def chain_relative_poses(relative_poses: np.ndarray) -> np.ndarray:
    """Sequential product of relative 3×3 transforms starting at the identity,
    reported as poses (modules/localization.py:111-130). -> (N+1, 3)."""
    T = T_from_pose([0.0, 0.0, 0.0])
    poses = [[0.0, 0.0, 0.0]]
    for Trel in relative_poses:
        T = T @ Trel
        poses.append(pose_from_T(T))
    return np.array(poses)


def poses_from_scan_matching(x_ts, z_ts, return_relative_poses=False, *, nn="kdtree",
                             recipe="f64", return_iterations=False):
    """modules/localization.py:95-134: for every consecutive scan pair, ICP with
    source = scan i, target = scan i-1, initial guess from the
    *odometry* poses only, epsilon = 0.001 and the other defaults; then the
    SE(2) chain product."""
    x_ts = np.asarray(x_ts, dtype=np.float64)
    rel = []
    iters = []
    for i in range(1, x_ts.shape[0]):
        T_init = get_relative_pose(x_ts[i - 1], x_ts[i])
        r = icp_oracle.run_icp_full(z_ts[i], z_ts[i - 1], init_transform=TSE3_from_TSE2(T_init),
                                    epsilon=0.001, nn=nn, recipe=recipe)
        rel.append(TSE2_from_TSE3(r.T))
        iters.append(r.iterations)
    rel = np.array(rel)
    poses = chain_relative_poses(rel)
    out = (poses,)
    if return_relative_poses:
        out += (rel,)
    if return_iterations:
        out += (np.array(iters),)
    return out if len(out) > 1 else out[0]


def lidar_xy_from_ranges(ranges, range_min, range_max):
    """Polar -> Cartesian robot-frame points of one 1081-beam scan, beams spread
    over [-135°, +135°], lidar offset (0.13323, 0) added
    (modules/localization.py:252-307)."""
    ranges = np.asarray(ranges, dtype=np.float64)
    ang = np.linspace(-135 * (np.pi / 180), 135 * (np.pi / 180), ranges.shape[0])
    ok = (ranges >= range_min) & (ranges <= range_max)
    r, a = ranges[ok], ang[ok]
    return np.column_stack((r * np.cos(a) + 0.13323, r * np.sin(a) + 0.0))
