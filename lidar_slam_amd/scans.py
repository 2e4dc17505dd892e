"""Device-resident scan sets: the scans of a trajectory kept in HBM between pipeline stages.

The reference passes the scans from stage to stage as a Python list (main.py:84-87 → :120 → :205).
``ScanSet`` keeps them on the GPU (float64 for ray casting, float32 for ICP, CSR offsets) and is
accepted wherever this package takes ``z_ts``: ``poses_from_scan_matching``,
``OccupancyGridMap.build_map``.  It still *reads* like the reference's list — ``len()``, indexing
and iteration download the points once, on first use.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import check, lib, ptr


class ScanSet:
    def __init__(self, handle: C.c_void_p, stats=None):
        self._handle = handle
        self.stats = stats                       # (mean, std, threshold) of the radial stage, if it ran
        K, total, max_n = C.c_int64(), C.c_int64(), C.c_int64()
        check(lib().lslam_scans_info(handle, C.byref(K), C.byref(total), C.byref(max_n)))
        self.n_scans, self.n_points, self.max_points = int(K.value), int(total.value), int(max_n.value)
        self._host = None

    # ---- construction -------------------------------------------------------------------------
    @classmethod
    def from_ranges(cls, lidar_ranges, lidar_range_min, lidar_range_max, dbscan=None, k_std=None, device=0):
        """get_lidar_data [→ DBSCAN_outliers_removal(eps, min_samples)] [→ statistical_outliers_removal(k_std)]
        (modules/localization.py:252-307, 201-221, 223-250), results kept on the device."""
        from .modules.localization import LIDAR_OFFSET, _ANGLE_MAX, _ANGLE_MIN
        ranges = np.ascontiguousarray(lidar_ranges, dtype=np.float64)
        K, B = ranges.shape
        angles = np.linspace(_ANGLE_MIN, _ANGLE_MAX, B)
        return cls._create(_lib.PRE_POLAR, ranges, K, B, np.cos(angles), np.sin(angles), LIDAR_OFFSET,
                           lidar_range_min, lidar_range_max, None, None, dbscan, k_std, device)

    @classmethod
    def from_scans(cls, z_ts, dbscan=None, k_std=None, device=0):
        """Upload a list of (n_i, 2) scans (optionally through the two outlier filters)."""
        from .modules.localization import _pack64
        xy, off = _pack64(z_ts)
        return cls._create(0, None, off.shape[0] - 1, 0, None, None, (0.0, 0.0), 0.0, 0.0, xy, off, dbscan, k_std, device)

    @classmethod
    def _create(cls, stages, ranges, K, B, cos_a, sin_a, offset, rmin, rmax, xy, off, dbscan, k_std, device):
        eps, min_samples = dbscan if dbscan is not None else (0.0, 0)
        if dbscan is not None:
            stages |= _lib.PRE_DBSCAN
        if k_std is not None:
            stages |= _lib.PRE_RADIAL
        stats = np.zeros(3, dtype=np.float64)
        handle = C.c_void_p()
        check(lib().lslam_scans_create(stages, ptr(ranges), K, B, ptr(cos_a), ptr(sin_a), offset[0], offset[1],
                                       float(rmin), float(rmax), ptr(xy), ptr(off), float(eps), int(min_samples),
                                       float(k_std if k_std is not None else 0.0), ptr(stats), int(device), C.byref(handle)))
        return cls(handle, tuple(stats) if k_std is not None else None)

    def __del__(self):
        h = getattr(self, "_handle", None)
        if h is not None and h.value:
            try:
                lib().lslam_scans_destroy(h)
            except Exception:
                pass
            self._handle = C.c_void_p()

    # ---- list-like reading (lazy download) -----------------------------------------------------
    def to_host(self):
        """-> PackedScans (a list of (n_i, 2) float64 views of one packed array)"""
        if self._host is None:
            from .batch import PackedScans
            xy = np.empty((max(self.n_points, 1), 2), dtype=np.float64)
            off = np.zeros(self.n_scans + 1, dtype=np.int64)
            check(lib().lslam_scans_download(self._handle, ptr(xy), ptr(off)))
            self._host = PackedScans(xy[: self.n_points], off)
        return self._host

    def __len__(self):
        return self.n_scans

    def __getitem__(self, i):
        return self.to_host()[i]

    def __iter__(self):
        return iter(self.to_host())

    # ---- device-side consumers -------------------------------------------------------------------
    def match_sequence(self, init_T, epsilon=0.001, max_iters=2000, stopping_thresh=0.0001, normalize_error=False):
        """ICP of every scan s+1 onto scan s (modules/localization.py:115-124), all on the device."""
        from .batch import IcpBatchResult
        S = max(self.n_scans - 1, 0)
        init_T = np.ascontiguousarray(init_T, dtype=np.float64).reshape(S, 4, 4)
        out_T = np.empty((S, 4, 4), dtype=np.float64)
        out_err = np.empty(S, dtype=np.float64)
        out_it = np.empty(S, dtype=np.int32)
        flags = _lib.ICP_NORMALIZE_ERROR if normalize_error else 0
        check(lib().lslam_scans_match_sequence(self._handle, ptr(init_T), float(epsilon), float(stopping_thresh),
                                               int(max_iters), flags, ptr(out_T), ptr(out_err), ptr(out_it)))
        return IcpBatchResult(out_T, out_err, out_it, None)
