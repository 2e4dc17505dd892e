"""NumPy restatement of the LiDAR pre-processing in front of the ICP path (test oracle).

TEST INFRASTRUCTURE — see ``oracle/__init__.py``.  Parity: PINNED by ``tests/golden/lidar_pre.npz``
(the reference's ``get_lidar_data`` → ``DBSCAN_outliers_removal(eps=0.1, min_samples=10)`` →
``statistical_outliers_removal(k_std=2)`` chain of main.py:84-87 run on eight variations of the
reference's real scan).
"""
from __future__ import annotations

import numpy as np


def lidar_xy(ranges, range_min, range_max):
    """modules/localization.py:252-307: beams over [-135°, 135°], ranges outside
    [range_min, range_max] dropped, polar → Cartesian, lidar→body offset (0.13323, 0) added.
    ranges: (K, B) -> list of K arrays (n_k, 2)."""
    ranges = np.asarray(ranges, dtype=np.float64)
    ang = np.linspace(-135 * (np.pi / 180), 135 * (np.pi / 180), ranges.shape[1])
    out = []
    for r in ranges:
        ok = (r >= range_min) & (r <= range_max)
        out.append(np.column_stack((r[ok] * np.cos(ang[ok]) + 0.13323, r[ok] * np.sin(ang[ok]) + 0.0)))
    return out


def dbscan_keep_mask(xy, eps, min_samples):
    """Which points sklearn's DBSCAN does NOT label −1 (modules/localization.py:201-221 keeps
    ``labels != -1``): core points (≥ min_samples points, itself included, within eps) and border
    points (within eps of a core point).  Neighbourhood test on the squared distance, d² ≤ eps²,
    accumulated as dx·dx then + dy·dy like sklearn's KD-tree ``rdist``."""
    xy = np.asarray(xy, dtype=np.float64)
    dx = xy[:, None, 0] - xy[None, :, 0]
    dy = xy[:, None, 1] - xy[None, :, 1]
    near = (dx * dx + dy * dy) <= eps * eps
    core = near.sum(axis=1) >= min_samples
    border = (near & core[None, :]).any(axis=1)
    return core | border


def dbscan_outliers_removal(z_ts, eps=0.5, min_samples=10):
    return [np.asarray(z)[dbscan_keep_mask(z, eps, min_samples)] for z in z_ts]


def radial_threshold(z_ts, k_std):
    """mean + k·std of the point ranges over ALL scans (modules/localization.py:235-239)."""
    d = np.sqrt(np.sum(np.vstack(z_ts) ** 2, axis=1))
    return np.mean(d) + k_std * np.std(d), np.mean(d), np.std(d)


def statistical_outliers_removal(z_ts, k_std=3):
    """keep the points whose range is < the global threshold (modules/localization.py:223-250)."""
    thr, _, _ = radial_threshold(z_ts, k_std)
    return [np.asarray(z)[np.sqrt(np.sum(np.asarray(z) ** 2, axis=1)) < thr] for z in z_ts]
