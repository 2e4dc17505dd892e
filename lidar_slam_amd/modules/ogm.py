"""Log-odds occupancy grid on MI355X behind the call surface of the reference's ``modules/ogm.py``.

``OccupancyGridMap`` keeps the reference's constructor, attributes and methods
(modules/ogm.py:7-43,45,66,87,102,126,149,190).  The float32 log-odds grid lives in HBM; the
NumPy mirrors ``grid_map_log_odds`` / ``grid_map`` are materialised lazily on attribute access
(``modules/texture_mapping.py:47,87`` reads them after ``build_map``).  No CPU fallback.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from .._lib import check, lib, ptr


class OccupancyGridMap:

    def __init__(self, resolution, world_map_max_x, world_map_max_y, world_map_min_x, world_map_min_y,
                 buffer=1., device=0):
        # geometry exactly as modules/ogm.py:30-38
        self.res = float(resolution)
        self.world_map_max_x = float(world_map_max_x)
        self.world_map_max_y = float(world_map_max_y)
        self.world_map_min_x = float(world_map_min_x)
        self.world_map_min_y = float(world_map_min_y)
        self.buffer = float(buffer)
        self.grid_map_width = int(np.ceil((world_map_max_x - world_map_min_x) / resolution + buffer))
        self.grid_map_height = int(np.ceil((world_map_max_y - world_map_min_y) / resolution + buffer))
        self._shape = (self.grid_map_width, self.grid_map_height)
        self.grid_map = np.zeros(self._shape, dtype=np.uint8)
        self._logodds_ratio = float(np.log(4.))       # modules/ogm.py:43
        self._clip = 20.0                               # modules/ogm.py:188
        self._host_logodds = np.zeros(self._shape, dtype=np.float32)
        self._host_stale = False
        self._handle = C.c_void_p()
        check(lib().lslam_ogm_create(self.res, self.world_map_min_x, self.world_map_min_y,
                                     self.grid_map_width, self.grid_map_height, self._logodds_ratio,
                                     self._clip, int(device), C.byref(self._handle)))

    def __del__(self):
        h = getattr(self, "_handle", None)
        if h is not None and h.value:
            try:
                lib().lslam_ogm_destroy(h)
            except Exception:
                pass
            self._handle = C.c_void_p()

    # -- attributes the reference exposes ------------------------------------------------------
    @property
    def logodds_ratio(self):
        return self._logodds_ratio

    @logodds_ratio.setter
    def logodds_ratio(self, v):
        self._logodds_ratio = float(v)
        check(lib().lslam_ogm_set_params(self._handle, self._logodds_ratio, self._clip))

    @property
    def grid_map_log_odds(self):
        """float32 (width, height) log-odds; a fresh array after every update, as the reference
        rebinds it in update_map (modules/ogm.py:188)."""
        if self._host_stale:
            out = np.empty(self._shape, dtype=np.float32)
            check(lib().lslam_ogm_read_logodds(self._handle, ptr(out)))
            self._host_logodds = out
            self._host_stale = False
        return self._host_logodds

    @grid_map_log_odds.setter
    def grid_map_log_odds(self, value):
        arr = np.ascontiguousarray(value, dtype=np.float32)
        if arr.shape != self._shape:
            raise ValueError(f"log-odds grid must have shape {self._shape}")
        check(lib().lslam_ogm_write_logodds(self._handle, ptr(arr)))
        self._host_logodds = arr
        self._host_stale = False

    # -- mapping --------------------------------------------------------------------------------
    def build_map(self, states, meas):
        """All scans applied in order, then the binarisation of modules/ogm.py:61-64
        (modules/ogm.py:45-64).  One library call for the whole trajectory."""
        states = np.ascontiguousarray(np.asarray(states, dtype=np.float64)[:, :3])
        K = states.shape[0]
        from ..batch import PackedScans
        from ..scans import ScanSet
        if isinstance(meas, ScanSet):                   # scans already in HBM
            if len(meas) != K:
                raise ValueError("need one scan per state")
            check(lib().lslam_ogm_update_scans(self._handle, meas._handle, ptr(states)))
            self._host_stale = True
            out = np.empty(self._shape, dtype=np.uint8)
            check(lib().lslam_ogm_binarize(self._handle, ptr(out)))
            self.grid_map = out
            return
        packed = meas.packed() if isinstance(meas, PackedScans) and len(meas) == K else None
        if packed is not None and packed[0].dtype == np.float64 and packed[0].ndim == 2 and packed[0].shape[1] == 2:
            flat, off = np.ascontiguousarray(packed[0]), packed[1]        # already packed: no copies
        else:
            staged = None
            if isinstance(meas, list) and len(meas) == K and K:
                # a plain list of (n_i, 2) arrays: packed by the C helper (a few threads) straight into the library's pinned
                # staging buffer, from where the upload is one DMA; anything it cannot take is packed with NumPy below
                from ..batch import _pack_staged
                try:
                    staged = _pack_staged(meas, np.float64, 0)
                except ValueError:
                    staged = None                                 # non-finite points: the reference does not check here either
            if staged is not None:
                flat, off = staged
            else:
                off = np.zeros(K + 1, dtype=np.int64)
                scans = [np.asarray(meas[i], dtype=np.float64).reshape(-1, 2) for i in range(K)]
                off[1:] = np.cumsum([s.shape[0] for s in scans])
                flat = np.ascontiguousarray(np.concatenate(scans, axis=0)) if K else np.zeros((0, 2))
        check(lib().lslam_ogm_update(self._handle, ptr(states), ptr(flat), ptr(off), K))
        self._host_stale = True
        out = np.empty(self._shape, dtype=np.uint8)
        check(lib().lslam_ogm_binarize(self._handle, ptr(out)))
        self.grid_map = out

    def update_map(self, x_t, z_t):
        """One scan (modules/ogm.py:149-188): rays from (x, y) + (0.13323, 0) to every rotated and
        translated beam end, −log 4 along the ray, +log 4 on its last in-map cell, clip to ±20."""
        pose = np.ascontiguousarray(np.asarray(x_t, dtype=np.float64).reshape(3))
        z = np.ascontiguousarray(np.asarray(z_t, dtype=np.float64).reshape(-1, 2))
        off = np.array([0, z.shape[0]], dtype=np.int64)
        check(lib().lslam_ogm_update(self._handle, ptr(pose), ptr(z), ptr(off), 1))
        self._host_stale = True

    def scan_counts(self, x_t, z_t):
        """int32 (width, height) grid of hits − misses ONE scan would add (not applied)."""
        pose = np.ascontiguousarray(np.asarray(x_t, dtype=np.float64).reshape(3))
        z = np.ascontiguousarray(np.asarray(z_t, dtype=np.float64).reshape(-1, 2))
        out = np.empty(self._shape, dtype=np.int32)
        check(lib().lslam_ogm_scan_counts(self._handle, ptr(pose), ptr(z), z.shape[0], ptr(out)))
        return out

    # -- index helpers (host, O(1) per point; modules/ogm.py:102-147) ---------------------------
    def world2grid(self, x, y):
        if isinstance(x, np.ndarray) and isinstance(y, np.ndarray):
            x_g = np.ceil((x - self.world_map_min_x) / self.res).astype(np.int32) - 1
            y_g = np.ceil((y - self.world_map_min_y) / self.res).astype(np.int32) - 1
            return np.hstack((x_g.reshape(-1, 1), y_g.reshape(-1, 1)))
        return np.array([int(np.ceil((x - self.world_map_min_x) / self.res)) - 1,
                         int(np.ceil((y - self.world_map_min_y) / self.res)) - 1], dtype=np.int32)

    def grid2world(self, i, j):
        if isinstance(i, np.ndarray) and isinstance(j, np.ndarray):
            return np.hstack(((i * self.res + self.world_map_min_x).reshape(-1, 1),
                              (j * self.res + self.world_map_min_y).reshape(-1, 1)))
        return np.array([i * self.res + self.world_map_min_x, j * self.res + self.world_map_min_y])

    def bresenham2D(self, sx, sy, ex, ey):
        """Cells of the integer line, both ends included (modules/ogm.py:190-231), from the same
        closed form the ray-casting kernel evaluates per cell."""
        sx, sy, ex, ey = int(sx), int(sy), int(ex), int(ey)
        dx, dy = abs(ex - sx), abs(ey - sy)
        steep = dy > dx
        if steep:
            dx, dy = dy, dx
        k = np.arange(dx + 1)
        minor = (k * dy + dx - 1 - dx // 2) // dx if dy else np.zeros_like(k)
        gx, gy = (1 if sx <= ex else -1), (1 if sy <= ey else -1)
        if steep:
            return np.vstack((sx + gx * minor, sy + gy * k)).T.astype(int)
        return np.vstack((sx + gx * k, sy + gy * minor)).T.astype(int)

    # -- plots (matplotlib file output, outside the accelerated path; modules/ogm.py:66-100) --------
    def plot_log_odds_map(self, fname):
        """Save the min-max normalised, square-root emphasised log-odds as a gray image."""
        import matplotlib.pyplot as plt
        lo = self.grid_map_log_odds
        span = np.max(lo) - np.min(lo)
        img = np.power((lo - np.min(lo)) / span, 1 / 2)
        plt.figure(figsize=(10, 10))
        plt.imshow(img, cmap="gray", interpolation="nearest")
        plt.savefig(fname)
        plt.close()

    def plot_map(self, fname):
        """Save the binarised map as a gray image."""
        import matplotlib.pyplot as plt
        plt.figure(figsize=(10, 10))
        plt.imshow(self.grid_map, cmap="gray")
        plt.savefig(fname)
        plt.close()
