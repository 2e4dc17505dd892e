"""Emits tests/golden/utils_glue.npz by RUNNING the reference's modules/utils.py glue functions (lines 140-240) on seeded
inputs.  Run in the build container only (the reference does not travel):
    PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg python3 tests/golden/make_golden_utils.py"""
import os
import sys

import numpy as np

sys.path.insert(0, "/root/reference")
import modules.utils as R  # noqa: E402

rng = np.random.default_rng(2024)
K = 12
poses = np.column_stack((rng.uniform(-30, 30, K), rng.uniform(-30, 30, K), rng.uniform(-np.pi, np.pi, K)))
T3 = np.array([R.T_from_pose(p) for p in poses])
T4 = []
for k in range(K):
    T = np.eye(4)
    a, b, c = rng.uniform(-np.pi, np.pi, 3)
    Rz = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]])
    Ry = np.array([[np.cos(b), 0, np.sin(b)], [0, 1, 0], [-np.sin(b), 0, np.cos(b)]])
    Rx = np.array([[1, 0, 0], [0, np.cos(c), -np.sin(c)], [0, np.sin(c), np.cos(c)]])
    T[:3, :3] = Rz @ Ry @ Rx
    T[:3, 3] = rng.uniform(-5, 5, 3)
    T4.append(T)
T4 = np.array(T4)
pts3 = rng.normal(size=(257, 3)) * 7.0
pts2 = rng.normal(size=(301, 2)) * 7.0
out = {
    "poses": poses, "T_from_pose": T3, "T4": T4, "pts3": pts3, "pts2": pts2,
    "pose_from_T3": np.array([R.pose_from_T(T) for T in T3]),
    "pose_from_T4": np.array([R.pose_from_T(T) for T in T4]),
    "TSE3_from_TSE2": np.array([R.TSE3_from_TSE2(T) for T in T3]),
    "TSE2_from_TSE3": np.array([R.TSE2_from_TSE3(T) for T in T4]),
    "relative": np.array([R.get_relative_pose(poses[k], poses[k + 1]) for k in range(K - 1)]),
    "pts3_T4": R.transform_points(pts3, T4[0]),
    "pts2_T3": R.transform_points(pts2, T3[1]),
}
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "utils_glue.npz"), **out)
print({k: v.shape for k, v in out.items()})
