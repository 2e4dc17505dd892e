#!/usr/bin/env python3
"""Emit tests/golden/icp_multistart.npz by RUNNING THE REFERENCE's run_icp over the 24 yaw starts of
code/icp_warm_up/test_icp.py:66-92 (the script itself needs open3d, which is absent, so the loop is
driven from here with the reference's run_icp and its argument values).  Build container only:

    PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg python3 tests/golden/make_golden_multistart.py

Inputs are the clouds of icp_3d.npz; stored: every start's pose / error / iteration count, and the
index, pose and error the reference's `err < best_error` scan keeps.
"""
import os
import sys

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REF)
sys.dont_write_bytecode = True

import modules.icp as ref_icp                      # noqa: E402

_count = {"n": 0}
_orig = ref_icp.icp_iteration


def _counting(*a, **k):
    _count["n"] += 1
    return _orig(*a, **k)


ref_icp.icp_iteration = _counting

g = np.load(os.path.join(HERE, "icp_3d.npz"))
src, tgt = g["src"], g["tgt"]
poses, errors, iters = [], [], []
best_error, best_pose, best_k = np.inf, None, -1
for k, yaw in enumerate(np.linspace(0, 2 * np.pi, 24, endpoint=False)):
    T = np.array([[np.cos(yaw), -np.sin(yaw), 0, 0], [np.sin(yaw), np.cos(yaw), 0, 0], [0, 0, 1, 0], [0, 0, 0, 1]])
    T[0:3, 3] = np.mean(tgt, axis=0) - T[0:3, 0:3] @ np.mean(src, axis=0)
    _count["n"] = 0
    pose, err, corr = ref_icp.run_icp(src, tgt, init_transform=T, epsilon=0.001, return_error=True,
                                      return_correspondences=True, normalize_error=True)
    poses.append(pose); errors.append(err); iters.append(_count["n"])
    if err < best_error:
        best_error, best_pose, best_k = err, pose, k
np.savez_compressed(os.path.join(HERE, "icp_multistart.npz"), T=np.array(poses), error=np.array(errors),
                    iterations=np.array(iters, dtype=np.int32), best=np.int64(best_k), best_pose=best_pose,
                    best_error=np.float64(best_error))
print("best start", best_k, "error", best_error, "iterations", iters)
