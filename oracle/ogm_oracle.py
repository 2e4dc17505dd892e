"""NumPy restatement of the log-odds occupancy-grid ray casting (test oracle).

TEST INFRASTRUCTURE — see ``oracle/__init__.py``.  Parity: PINNED by
``tests/golden/ogm_*.npz`` (emitted by the imported reference) and by the
reference's own Bresenham known-answer rays (``code/pr2_utils.py:88-100``).

Two update paths are offered:

* ``update_map_sequential`` – the reference's arithmetic, ray by ray in float32
  with a float64 step (modules/ogm.py:172-188);
* ``scan_counts`` + ``apply_counts`` – the device formulation: a per-scan int32
  grid of (hits − misses), then ``l ← clip(l + cnt·log 4, ±20)`` per touched cell
  in EXACT arithmetic: a cell is ``(base, n)`` with value ``base + n·log 4``; a scan
  adds its count to ``n``; only a clip (or an assignment of the grid) moves the base.
  ``grid_map_log_odds`` is ``f32(base + n·log 4)`` — one float64 rounding however
  many scans were applied.  The integer grid is the bit-exact contract; the float
  log-odds agree with the sequential path to a few float32 ulps.
"""
from __future__ import annotations

import numpy as np

LOG_ODDS_STEP = np.log(4.0)          # modules/ogm.py:43
LOG_ODDS_CLIP = 20.0                 # modules/ogm.py:188
LIDAR_OFFSET_XY = (0.13323, 0.0)     # modules/ogm.py:165 (p_rl[:2])


def bresenham_cells(sx: int, sy: int, ex: int, ey: int) -> np.ndarray:
    """Cells of the integer line (sx,sy)→(ex,ey), both ends included, one cell
    per unit of the major axis (modules/ogm.py:190-231).

    Restated through the error sequence the reference builds with
    ``mod(arange(floor(dx/2), ...,-dy), dx)``: e_k = (⌊dx/2⌋ − k·dy) mod dx and
    the minor axis advances at step k iff e_k ≥ e_{k−1}."""
    sx, sy, ex, ey = int(sx), int(sy), int(ex), int(ey)
    dx, dy = abs(ex - sx), abs(ey - sy)
    steep = dy > dx
    if steep:
        dx, dy = dy, dx
    minor = np.zeros(dx + 1, dtype=np.int64)
    if dy != 0:
        h = dx // 2
        e_prev = h % dx
        off = 0
        for k in range(1, dx + 1):
            e = (h - k * dy) % dx
            if e >= e_prev:
                off += 1
            minor[k] = off
            e_prev = e
    major = np.arange(dx + 1, dtype=np.int64)
    if steep:
        y = sy + major if sy <= ey else sy - major
        x = sx + minor if sx <= ex else sx - minor
    else:
        x = sx + major if sx <= ex else sx - major
        y = sy + minor if sy <= ey else sy - minor
    return np.stack((x, y), axis=1)


def bresenham_cells_closed_form(sx: int, sy: int, ex: int, ey: int) -> np.ndarray:
    """Same cells, every cell computed independently (what the HIP kernel does):
    minor offset of cell k = (k·dy + dx − 1 − ⌊dx/2⌋) // dx  (0 when dy == 0)."""
    sx, sy, ex, ey = int(sx), int(sy), int(ex), int(ey)
    dx, dy = abs(ex - sx), abs(ey - sy)
    steep = dy > dx
    if steep:
        dx, dy = dy, dx
    k = np.arange(dx + 1, dtype=np.int64)
    minor = (k * dy + dx - 1 - dx // 2) // dx if dy != 0 else np.zeros_like(k)
    sgx = 1 if sx <= ex else -1
    sgy = 1 if sy <= ey else -1
    if steep:
        return np.stack((sx + sgx * minor, sy + sgy * k), axis=1)
    return np.stack((sx + sgx * k, sy + sgy * minor), axis=1)


class OccupancyGridOracle:
    """Grid container with the reference's geometry (modules/ogm.py:7-43)."""

    def __init__(self, resolution, world_map_max_x, world_map_max_y, world_map_min_x,
                 world_map_min_y, buffer=1.0):
        self.res = float(resolution)
        self.world_map_max_x = float(world_map_max_x)
        self.world_map_max_y = float(world_map_max_y)
        self.world_map_min_x = float(world_map_min_x)
        self.world_map_min_y = float(world_map_min_y)
        self.buffer = float(buffer)
        self.grid_map_width = int(np.ceil((world_map_max_x - world_map_min_x) / resolution + buffer))
        self.grid_map_height = int(np.ceil((world_map_max_y - world_map_min_y) / resolution + buffer))
        self.grid_map = np.zeros((self.grid_map_width, self.grid_map_height), dtype=np.uint8)
        self.grid_map_log_odds = np.zeros((self.grid_map_width, self.grid_map_height), dtype=np.float32)
        self.logodds_ratio = LOG_ODDS_STEP

    # the log-odds are kept as (float32 base, int64 steps on it); reading rounds once, assigning resets the steps
    @property
    def grid_map_log_odds(self):
        return (self._base.astype(np.float64) + self._steps.astype(np.float64) * LOG_ODDS_STEP).astype(np.float32)

    @grid_map_log_odds.setter
    def grid_map_log_odds(self, value):
        self._base = np.array(value, dtype=np.float32)
        self._steps = np.zeros(self._base.shape, dtype=np.int64)

    # -- geometry ----------------------------------------------------------
    def world2grid(self, x, y):
        """ceil((v − min)/res) − 1 per axis in float64, then int32
        (modules/ogm.py:102-124); world2grid(min) = −1."""
        if isinstance(x, np.ndarray) and isinstance(y, np.ndarray):
            gx = np.ceil((x - self.world_map_min_x) / self.res).astype(np.int32) - 1
            gy = np.ceil((y - self.world_map_min_y) / self.res).astype(np.int32) - 1
            return np.stack((gx.ravel(), gy.ravel()), axis=1)
        return np.array([int(np.ceil((x - self.world_map_min_x) / self.res)) - 1,
                         int(np.ceil((y - self.world_map_min_y) / self.res)) - 1], dtype=np.int32)

    def grid2world(self, i, j):
        """i·res + min (modules/ogm.py:126-147)."""
        if isinstance(i, np.ndarray) and isinstance(j, np.ndarray):
            return np.stack(((i * self.res + self.world_map_min_x).ravel(),
                             (j * self.res + self.world_map_min_y).ravel()), axis=1)
        return np.array([i * self.res + self.world_map_min_x, j * self.res + self.world_map_min_y])

    def ray_endpoints(self, x_t, z_t):
        """Grid cell of the ray origin and of every beam end point
        (modules/ogm.py:160-170).  The scan is rotated about the robot origin and
        translated; the ray origin is (x,y) + (0.13323, 0) *un-rotated* (O1)."""
        x, y, yaw = (float(v) for v in x_t)
        c, s = np.cos(yaw), np.sin(yaw)
        R = np.array([[c, -s], [s, c]])
        zw = np.dot(np.asarray(z_t, dtype=np.float64), R.T) + np.array([x, y])
        start = self.world2grid(x + LIDAR_OFFSET_XY[0], y + LIDAR_OFFSET_XY[1])
        ends = self.world2grid(zw[:, 0], zw[:, 1])
        return start, ends

    def _valid_cells(self, start, end):
        cells = bresenham_cells_closed_form(start[0], start[1], end[0], end[1])
        ok = ((cells[:, 0] >= 0) & (cells[:, 0] < self.grid_map_width)
              & (cells[:, 1] >= 0) & (cells[:, 1] < self.grid_map_height))
        return cells[ok]                     # filtered *after* tracing (O2)

    # -- device formulation ------------------------------------------------
    def scan_counts(self, x_t, z_t) -> np.ndarray:
        """int32 grid of (hits − misses) of one scan: −1 on every valid cell of a
        ray but its last valid one, +1 on the last valid cell (O2, O3)."""
        cnt = np.zeros((self.grid_map_width, self.grid_map_height), dtype=np.int32)
        start, ends = self.ray_endpoints(x_t, z_t)
        for e in ends:
            cells = self._valid_cells(start, e)
            if cells.shape[0] == 0:
                continue
            np.subtract.at(cnt, (cells[:-1, 0], cells[:-1, 1]), 1)
            cnt[cells[-1, 0], cells[-1, 1]] += 1
        return cnt

    def apply_counts(self, cnt: np.ndarray) -> None:
        """one scan: n += cnt on the touched cells, then the clip of modules/ogm.py:188 as integer thresholds —
        base + n·step > clip  ⇔  n > ⌊(clip − base)/step⌋ (and symmetrically below)."""
        base, n = self._base, self._steps
        b64 = base.astype(np.float64)
        hi = np.floor((LOG_ODDS_CLIP - b64) / LOG_ODDS_STEP)
        lo = np.ceil((-LOG_ODDS_CLIP - b64) / LOG_ODDS_STEP)
        touched = cnt != 0
        n[touched] += cnt[touched]
        above = touched & (n > hi)
        below = touched & ~above & (n < lo)
        base[above] = LOG_ODDS_CLIP; n[above] = 0
        base[below] = -LOG_ODDS_CLIP; n[below] = 0

    def update_map(self, x_t, z_t) -> None:
        self.apply_counts(self.scan_counts(x_t, z_t))

    # -- reference arithmetic ----------------------------------------------
    def update_map_sequential(self, x_t, z_t) -> None:
        """Ray-by-ray float32 accumulation with a float64 step, clip once per scan
        (modules/ogm.py:172-188)."""
        g = self.grid_map_log_odds                      # a float32 copy; assigned back (steps reset) below
        start, ends = self.ray_endpoints(x_t, z_t)
        for e in ends:
            cells = self._valid_cells(start, e)
            if cells.shape[0] == 0:
                continue
            ix, iy = cells[:-1, 0], cells[:-1, 1]
            g[ix, iy] = (g[ix, iy].astype(np.float64) - LOG_ODDS_STEP).astype(np.float32)
            jx, jy = cells[-1]
            g[jx, jy] = np.float32(np.float64(g[jx, jy]) + LOG_ODDS_STEP)
        self.grid_map_log_odds = np.clip(g, -LOG_ODDS_CLIP, LOG_ODDS_CLIP)

    def binarize(self) -> None:
        """pmf = 1/(1+exp(l)) in float32; cell = 1 where pmf > 0.5 else 0
        (modules/ogm.py:61-64; the ``== 0.5 → 0.5`` store truncates to 0 in uint8)."""
        pmf = 1.0 / (1 + np.exp(self.grid_map_log_odds))
        self.grid_map[pmf > 0.5] = 1
        self.grid_map[pmf <= 0.5] = 0

    def build_map(self, states, meas, sequential=False) -> None:
        """modules/ogm.py:45-64."""
        step = self.update_map_sequential if sequential else self.update_map
        for x_t, z_t in zip(states, meas):
            step(x_t, z_t)
        self.binarize()
