"""Seeded synthetic workloads of BASELINE.json's configs.

The reference ships no data (``data/*.npz`` is git-ignored there), so benchmarks and the
full-size property tests run on these.  Everything is deterministic in its seed.

  pair_c2      one planar pair, N = M = 1081: a 1081-beam ±135° scan of a rectilinear room
  sequence_c3  dataset-20-like sequence: 4956 scans along a smooth random walk through a floor plan
  batch_c4     S pairs × N = M planar points, per-pair random SE(2) + 1 cm noise
  cloud_c5     dense 3-D pair on a closed surface (superellipsoid, ~0.1 m), N = M = 100 000
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

N_BEAMS = 1081
FOV = np.deg2rad(135.0)
LIDAR_OFFSET_X = 0.13323          # modules/localization.py:297


@dataclass
class PairBatch:
    src: np.ndarray        # float32 (Σn, d)
    src_off: np.ndarray    # int64 (S+1,)
    tgt: np.ndarray
    tgt_off: np.ndarray
    init_T: np.ndarray     # float64 (S,4,4)
    true_T: np.ndarray     # float64 (S,4,4): maps source frame onto target frame


def se2_to_T4(x, y, th):
    x, y, th = np.broadcast_arrays(np.asarray(x, float), np.asarray(y, float), np.asarray(th, float))
    T = np.zeros(x.shape + (4, 4))
    c, s = np.cos(th), np.sin(th)
    T[..., 0, 0] = c; T[..., 0, 1] = -s; T[..., 1, 0] = s; T[..., 1, 1] = c
    T[..., 0, 3] = x; T[..., 1, 3] = y
    T[..., 2, 2] = 1.0; T[..., 3, 3] = 1.0
    return T


# ------------------------------------------------------------------------------------------- rooms
def _segments_of_boxes(boxes):
    """axis-aligned boxes (x0,y0,x1,y1) -> wall segments (K,4) as (ax,ay,bx,by)"""
    seg = []
    for x0, y0, x1, y1 in boxes:
        seg += [(x0, y0, x1, y0), (x1, y0, x1, y1), (x1, y1, x0, y1), (x0, y1, x0, y0)]
    return np.array(seg, dtype=np.float64)


def floor_plan(seed: int, half=12.0, n_obstacles=14):
    """outer walls + interior rectangular obstacles, axis aligned"""
    rng = np.random.default_rng(seed)
    boxes = [(-half, -half, half, half)]
    for _ in range(n_obstacles):
        cx, cy = rng.uniform(-half + 2, half - 2, 2)
        if abs(cx) < 1.5 and abs(cy) < 1.5:           # keep the start area free
            continue
        w, h = rng.uniform(0.4, 3.0, 2)
        boxes.append((cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2))
    return _segments_of_boxes(boxes), boxes


def raycast(origins, headings, segs, angles, max_range=30.0):
    """ranges (P, B) of rays from origins (P,2) at world angle heading+angles to wall segments"""
    P = origins.shape[0]
    out = np.full((P, angles.shape[0]), max_range)
    ax, ay, bx, by = segs.T
    ex, ey = bx - ax, by - ay
    for s in range(0, P, 64):
        o = origins[s:s + 64]
        th = headings[s:s + 64, None] + angles[None, :]
        dx, dy = np.cos(th)[..., None], np.sin(th)[..., None]                  # (p,B,1)
        wx, wy = ax[None, None, :] - o[:, 0, None, None], ay[None, None, :] - o[:, 1, None, None]
        den = dx * ey - dy * ex
        with np.errstate(divide="ignore", invalid="ignore"):
            t = (wx * ey - wy * ex) / den            # along the ray
            u = (wx * dy - wy * dx) / den            # along the segment
        ok = (np.abs(den) > 1e-12) & (t > 1e-6) & (u >= 0.0) & (u <= 1.0)
        t = np.where(ok, t, max_range)
        out[s:s + 64] = np.minimum(t.min(axis=2), max_range)
    return out


def scan_points(ranges, angles, rmin=0.1, rmax=30.0):
    """polar -> robot-frame xy with the lidar offset, invalid ranges dropped
    (what modules/localization.py:252-307 produces)"""
    ok = (ranges >= rmin) & (ranges <= rmax)
    r, a = ranges[ok], angles[ok]
    return np.column_stack((r * np.cos(a) + LIDAR_OFFSET_X, r * np.sin(a)))


# ---------------------------------------------------------------------------------------------- C2
def pair_c2(seed: int = 0):
    """single planar pair, N = M = 1081; returns (src (1081,2), tgt (1081,2), T_true 4x4) float64"""
    rng = np.random.default_rng(seed)
    hw = rng.uniform(2.0, 10.0, 4)                                  # wall distances -x,+x,-y,+y
    segs = _segments_of_boxes([(-hw[0], -hw[2], hw[1], hw[3])])
    angles = np.linspace(-FOV, FOV, N_BEAMS)
    r = raycast(np.zeros((1, 2)), np.zeros(1), segs, angles)[0]
    tgt = np.column_stack((r * np.cos(angles), r * np.sin(angles))) + rng.normal(0, 0.005, (N_BEAMS, 2))
    T_true = se2_to_T4(0.05, 0.01, 0.03)
    Ti = np.linalg.inv(T_true)
    src = tgt @ Ti[:2, :2].T + Ti[:2, 3] + rng.normal(0, 0.005, (N_BEAMS, 2))
    return src, tgt, T_true


# ---------------------------------------------------------------------------------------------- C3
@dataclass
class ScanSequence:
    x_ts: np.ndarray        # (N,3) odometry poses (truth + drift)
    true_poses: np.ndarray  # (N,3)
    z_ts: list              # N arrays (n_i, 2)
    ranges: np.ndarray = None   # (N, 1081) raw ranges, dropped beams = 0 (below range_min)


def sequence_c3(n_scans: int = 4956, seed: int = 20, drop=(0.0, 0.4), half: float = 12.0,
                n_obstacles: int = 14) -> ScanSequence:
    """dataset-20-like: per-step motion inside the measured envelope (|Δθ| ≤ 0.12 rad, |t| ≤ 0.4 m),
    1081 beams over ±135°, 5 mm range noise, a random share of beams dropped per scan
    (stands for --filter_lidar; 650–1081 points per scan), odometry = truth + slow drift."""
    rng = np.random.default_rng(seed)
    segs, boxes = floor_plan(seed, half=half, n_obstacles=n_obstacles)
    angles = np.linspace(-FOV, FOV, N_BEAMS)
    poses = np.zeros((n_scans, 3))
    lim = half - 1.5
    v, w = 0.0, 0.0
    for i in range(1, n_scans):
        v = np.clip(0.97 * v + rng.normal(0, 0.004), 0.0, 0.03)                # m per 25 ms step
        w = np.clip(0.95 * w + rng.normal(0, 0.003), -0.02, 0.02)
        x, y, th = poses[i - 1]
        nx, ny = x + v * np.cos(th), y + v * np.sin(th)
        blocked = abs(nx) > lim or abs(ny) > lim or any(
            b[0] - 0.6 < nx < b[2] + 0.6 and b[1] - 0.6 < ny < b[3] + 0.6 for b in boxes[1:])
        if blocked:                                                           # turn in place
            nx, ny, w = x, y, 0.02
        poses[i] = (nx, ny, th + w)
    lidar_xy = poses[:, :2] + LIDAR_OFFSET_X * np.column_stack((np.cos(poses[:, 2]), np.sin(poses[:, 2])))
    ranges = raycast(lidar_xy, poses[:, 2], segs, angles) + rng.normal(0, 0.005, (n_scans, N_BEAMS))
    z_ts = []
    for i in range(n_scans):
        keep = rng.random(N_BEAMS) >= rng.uniform(*drop)
        ranges[i] = np.where(keep, ranges[i], 0.0)
        z_ts.append(scan_points(ranges[i], angles))
    drift = np.cumsum(rng.normal(0, [0.0008, 0.0008, 0.0005], (n_scans, 3)), axis=0)
    drift[0] = 0
    return ScanSequence(x_ts=poses + drift, true_poses=poses, z_ts=z_ts, ranges=ranges)


# ---------------------------------------------------------------------------------------------- C4
def batch_c4(S: int = 4096, N: int = 4096, seed: int = 0, init_noise=(0.005, 0.003)) -> PairBatch:
    """S pairs, target ~ U(−10,10)², source = random SE(2) (|t| ≤ 0.3, |θ| ≤ 0.1) of it + N(0, 1 cm);
    the stream of pair s is seeded by (seed, s).  init = truth + N(5 mm, 3 mrad)."""
    src = np.empty((S * N, 2), dtype=np.float32)
    tgt = np.empty((S * N, 2), dtype=np.float32)
    true_T = np.empty((S, 4, 4))
    init_T = np.empty((S, 4, 4))
    for s in range(S):
        rng = np.random.default_rng([seed, s])
        q = rng.uniform(-10, 10, (N, 2))
        th = rng.uniform(-0.1, 0.1)
        t = rng.uniform(-0.3, 0.3, 2)
        T = se2_to_T4(t[0], t[1], th)
        Ti = np.linalg.inv(T)
        p = q @ Ti[:2, :2].T + Ti[:2, 3] + rng.normal(0, 0.01, (N, 2))
        src[s * N:(s + 1) * N] = p
        tgt[s * N:(s + 1) * N] = q
        true_T[s] = T
        init_T[s] = se2_to_T4(t[0] + rng.normal(0, init_noise[0]), t[1] + rng.normal(0, init_noise[0]),
                              th + rng.normal(0, init_noise[1]))
    off = np.arange(S + 1, dtype=np.int64) * N
    return PairBatch(src, off, tgt, off.copy(), init_T, true_T)


# ---------------------------------------------------------------------------------------------- C5
def _superellipsoid(rng, n, radii=(0.10, 0.06, 0.08), e=0.7):
    u = rng.uniform(-np.pi / 2, np.pi / 2, n)
    v = rng.uniform(-np.pi, np.pi, n)
    f = lambda w: np.sign(w) * np.abs(w) ** e           # noqa: E731
    return np.column_stack((radii[0] * f(np.cos(u)) * f(np.cos(v)),
                            radii[1] * f(np.cos(u)) * f(np.sin(v)),
                            radii[2] * f(np.sin(u))))


def cloud_c5(N: int = 100_000, M: int = 100_000, seed: int = 5):
    """dense 3-D pair: two independent samplings of one closed surface, 0.3 mm noise, the source
    moved by a small rigid motion.  Returns (src float32 (N,3), tgt float32 (M,3), T_true, T_init)."""
    rng = np.random.default_rng(seed)
    tgt = _superellipsoid(rng, M) + rng.normal(0, 0.0003, (M, 3))
    p = _superellipsoid(rng, N) + rng.normal(0, 0.0003, (N, 3))
    ax = np.array([0.2, -0.5, 0.8]); ax /= np.linalg.norm(ax)
    ang = 0.06
    K = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
    R = np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * K @ K
    T = np.eye(4); T[:3, :3] = R; T[:3, 3] = (0.004, -0.003, 0.002)
    Ti = np.linalg.inv(T)
    src = p @ Ti[:3, :3].T + Ti[:3, 3]
    return src.astype(np.float32), tgt.astype(np.float32), T, np.eye(4)


def _drill_surface(rng, n):
    """points on a drill-like object, metres: a box (body), a cylinder under it (handle), a thin cylinder in
    front (bit) — no symmetry, so only one of the yaw starts lands in the right basin"""
    k = n // 3
    f = rng.integers(0, 3, k); u, v = rng.uniform(-1, 1, k), rng.uniform(-1, 1, k); sgn = rng.choice([-1.0, 1.0], k)
    half = np.array([0.09, 0.03, 0.035])
    body = np.where((f == 0)[:, None], np.column_stack((sgn, u, v)), np.where((f == 1)[:, None], np.column_stack((u, sgn, v)), np.column_stack((u, v, sgn)))) * half
    a, z = rng.uniform(0, 2 * np.pi, k), rng.uniform(-0.13, -0.035, k)
    handle = np.column_stack((-0.04 + 0.018 * np.cos(a), 0.018 * np.sin(a), z))
    a, x = rng.uniform(0, 2 * np.pi, n - 2 * k), rng.uniform(0.09, 0.16, n - 2 * k)
    bit = np.column_stack((x, 0.006 * np.cos(a), 0.006 * np.sin(a)))
    return np.vstack((body, handle, bit))


def drill_c1(n_model: int = 6000, n_view: int = 4000, n_targets: int = 4, seed: int = 1):
    """config C1 (code/icp_warm_up/test_icp.py): a canonical object model and `n_targets` measured partial views of
    it, each seen from another yaw (any of the 24 yaw starts may be the good one) with a small tilt and offset.
    Returns (model (n_model,3) f64, [views (≈n_view,3) f64], [T_true model→view])."""
    rng = np.random.default_rng(seed)
    model = _drill_surface(rng, n_model)
    views, truths = [], []
    for k in range(n_targets):
        yaw, tilt = rng.uniform(0, 2 * np.pi), rng.uniform(-0.1, 0.1)
        Rz = np.array([[np.cos(yaw), -np.sin(yaw), 0], [np.sin(yaw), np.cos(yaw), 0], [0, 0, 1]])
        Rx = np.array([[1, 0, 0], [0, np.cos(tilt), -np.sin(tilt)], [0, np.sin(tilt), np.cos(tilt)]])
        R, t = Rz @ Rx, rng.uniform(-0.3, 0.3, 3)
        pts = _drill_surface(rng, 2 * n_view)
        pts = pts[pts @ np.array([0.3, 0.8, 0.52]) > -0.045][:n_view]          # the sensor misses the far side of the object
        views.append(pts @ R.T + t + rng.normal(0, 0.0005, pts.shape))
        T = np.eye(4); T[:3, :3] = R; T[:3, 3] = t
        truths.append(T)
    return model, views, truths
