#!/usr/bin/env python3
"""Emit the golden vectors under tests/golden/ by RUNNING THE REFERENCE.

Runs only in the build container (needs /root/reference); the GPU box never sees
the reference, only the .npz files written here.  Usage:

    PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg python3 tests/golden/make_golden.py

What is captured (SURVEY §8(c) G1–G8):
  icp_planar.npz    G1  8 planar scan pairs built from code/test_ranges.npy, eps=1e-3
  icp_forced50.npz  G2  2 of them with exactly 50 iterations (eps=0, thresh=0, max_iters=49)
  icp_loopclose.npz G3  identity init, normalize_error=True, default eps (main.py:162 style)
  icp_3d.npz        G4  synthetic 3-D object, partial view, 4 of the 24 yaw starts of test_icp.py:70-88
  icp_multistart.npz G4' all 24 yaw starts on the clouds of icp_3d.npz + the best-of pick (make_golden_multistart.py)
  kabsch.npz        G5  get_transform alone: planar (incl. reflections), 3-D, single point
  seq_scanmatch.npz G6  32-scan poses_from_scan_matching sequence
  ogm_main.npz      G7  main.py grid geometry (res .05, ±30 m): integer hit/miss grids + float log-odds
  ogm_small.npz     G7' 12 m map that most rays leave (O2/O3) + binarised map
  voxel.npz         G8  voxel_downsample on 1000 random points
  se2_outputs20.npz     first 512 entries of the reference's committed outputs/*.npy
  test_ranges.npy       the reference's one real Hokuyo scan (data file of code/pr2_utils.py tests)
"""
import os
import shutil
import sys

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REF)
sys.dont_write_bytecode = True
os.environ.setdefault("MPLBACKEND", "Agg")

import modules.icp as ref_icp                      # noqa: E402
import modules.localization as ref_loc             # noqa: E402
import modules.ogm as ref_ogm                      # noqa: E402
import modules.utils as ref_utils                  # noqa: E402


def save(name, **arrays):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **arrays)
    print(f"{name:22s} {os.path.getsize(path) / 1024:8.1f} KiB")


# ---------------------------------------------------------------- helpers
_iter_counter = {"n": 0}
_orig_iteration = ref_icp.icp_iteration


def _counting_iteration(*a, **k):
    _iter_counter["n"] += 1
    return _orig_iteration(*a, **k)


ref_icp.icp_iteration = _counting_iteration     # run_icp looks the name up at call time


def ref_run(src, tgt, **kw):
    """reference run_icp with all outputs + the number of iterations executed"""
    _iter_counter["n"] = 0
    T, err, corr = ref_icp.run_icp(src, tgt, return_error=True, return_correspondences=True, **kw)
    return T, float(err), np.asarray(corr, dtype=np.int64), _iter_counter["n"]


def se2(x, y, th):
    c, s = np.cos(th), np.sin(th)
    return np.array([[c, -s, x], [s, c, y], [0, 0, 1.0]])


def apply_se2(T, pts):
    return pts @ T[:2, :2].T + T[:2, 2]


def ragged(list_of_arrays):
    off = np.zeros(len(list_of_arrays) + 1, dtype=np.int64)
    off[1:] = np.cumsum([a.shape[0] for a in list_of_arrays])
    return np.concatenate(list_of_arrays, axis=0), off


# ------------------------------------------------------------------ data
ranges = np.load(os.path.join(REF, "code", "test_ranges.npy"))
shutil.copyfile(os.path.join(REF, "code", "test_ranges.npy"), os.path.join(HERE, "test_ranges.npy"))
scan = ref_loc.get_lidar_data(ranges[None, :], 0.1, 30)[0]        # (1076, 2)
print("real scan:", scan.shape)


def make_pair(rng, max_t=0.4, max_th=0.12, init_sigma=(0.005, 0.003)):
    """target = scan; source = the same world seen after a motion inside the dataset-20
    envelope, 5 mm noise, 5 % of the beams dropped; init = truth + small error."""
    th = rng.uniform(-max_th, max_th)
    t = rng.uniform(-max_t, max_t, size=2) / np.sqrt(2)
    T_true = se2(t[0], t[1], th)                    # maps source frame -> target frame
    keep = rng.random(scan.shape[0]) > 0.05
    src = apply_se2(np.linalg.inv(T_true), scan[keep]) + rng.normal(0, 0.005, (keep.sum(), 2))
    T_init = se2(t[0] + rng.normal(0, init_sigma[0]), t[1] + rng.normal(0, init_sigma[0]),
                 th + rng.normal(0, init_sigma[1]))
    return src, scan.copy(), ref_utils.TSE3_from_TSE2(T_init), T_true


# ------------------------------------------------------------- G1 / G2 / G3
rng = np.random.default_rng(20240807)
srcs, inits, Ts, errs, corrs, iters, trues = [], [], [], [], [], [], []
for _ in range(8):
    src, tgt, T0, T_true = make_pair(rng)
    T, err, corr, n = ref_run(src, tgt, init_transform=T0, epsilon=0.001)
    srcs.append(src); inits.append(T0); Ts.append(T); errs.append(err); corrs.append(corr)
    iters.append(n); trues.append(T_true)
src_cat, src_off = ragged(srcs)
corr_cat, _ = ragged([c[:, None] for c in corrs])
save("icp_planar.npz", src=src_cat, src_off=src_off, tgt=scan, init=np.array(inits), T=np.array(Ts),
     error=np.array(errs), corr=corr_cat[:, 0], iterations=np.array(iters), T_true=np.array(trues),
     epsilon=0.001)
print("  G1 iterations:", iters)

f_T, f_err, f_corr, f_it = [], [], [], []
for k in range(2):
    T, err, corr, n = ref_run(srcs[k], scan, init_transform=inits[k], epsilon=0.0, stopping_thresh=0.0,
                              max_iters=49)
    f_T.append(T); f_err.append(err); f_corr.append(corr); f_it.append(n)
save("icp_forced50.npz", src=np.concatenate(srcs[:2]), src_off=src_off[:3], tgt=scan,
     init=np.array(inits[:2]), T=np.array(f_T), error=np.array(f_err),
     corr=np.concatenate(f_corr), iterations=np.array(f_it))
print("  G2 iterations:", f_it)

rng = np.random.default_rng(162)
l_src, l_T, l_err, l_corr, l_it, l_eps = [], [], [], [], [], []
for k in range(4):
    src, tgt, _, _ = make_pair(rng, max_t=0.25, max_th=0.08)
    eps = 0.01 if k < 2 else 2e-5          # default eps fires at once; a tighter one must iterate
    T, err, corr, n = ref_run(src, tgt, normalize_error=True, epsilon=eps)      # identity init
    l_src.append(src); l_T.append(T); l_err.append(err); l_corr.append(corr); l_it.append(n); l_eps.append(eps)
cat, off = ragged(l_src)
save("icp_loopclose.npz", src=cat, src_off=off, tgt=scan, T=np.array(l_T), error=np.array(l_err),
     corr=np.concatenate(l_corr), iterations=np.array(l_it), epsilon=np.array(l_eps))
print("  G3 iterations:", l_it, "errors:", np.round(l_err, 6))

# --------------------------------------------------------------------- G4
rng = np.random.default_rng(4)


def drill_like(n):
    """cylinder (handle) + box (body) + thin cylinder (bit), metres"""
    k = n // 3
    a = rng.uniform(0, 2 * np.pi, k); z = rng.uniform(-0.10, 0.02, k)
    handle = np.column_stack((0.02 * np.cos(a), 0.02 * np.sin(a) - 0.01, z))
    face = rng.integers(0, 3, k); u = rng.uniform(-1, 1, (k, 3)); sgn = rng.choice([-1.0, 1.0], k)
    u[np.arange(k), face] = sgn
    body = u * np.array([0.06, 0.025, 0.03]) + np.array([0.01, 0, 0.05])
    a = rng.uniform(0, 2 * np.pi, n - 2 * k); x = rng.uniform(0.07, 0.13, n - 2 * k)
    bit = np.column_stack((x, 0.005 * np.cos(a), 0.05 + 0.005 * np.sin(a)))
    return np.vstack((handle, body, bit))


model = drill_like(1500)
yaw_t, tilt = 1.1, 0.2
Rz = np.array([[np.cos(yaw_t), -np.sin(yaw_t), 0], [np.sin(yaw_t), np.cos(yaw_t), 0], [0, 0, 1]])
Rx = np.array([[1, 0, 0], [0, np.cos(tilt), -np.sin(tilt)], [0, np.sin(tilt), np.cos(tilt)]])
world = model @ (Rz @ Rx).T + np.array([0.6, 0.1, 0.05])
view = world[world[:, 0] < np.quantile(world[:, 0], 0.7)]            # partial view
target3 = view + rng.normal(0, 0.0005, view.shape)
g_init, g_T, g_err, g_corr, g_it = [], [], [], [], []
for yaw in np.linspace(0, 2 * np.pi, 24, endpoint=False)[[0, 4, 5, 13]]:
    T0 = np.eye(4)
    T0[:3, :3] = np.array([[np.cos(yaw), -np.sin(yaw), 0], [np.sin(yaw), np.cos(yaw), 0], [0, 0, 1]])
    T0[:3, 3] = target3.mean(axis=0) - T0[:3, :3] @ model.mean(axis=0)      # test_icp.py:71-74
    T, err, corr, n = ref_run(model, target3, init_transform=T0, epsilon=0.001, normalize_error=True)
    g_init.append(T0); g_T.append(T); g_err.append(err); g_corr.append(corr); g_it.append(n)
save("icp_3d.npz", src=model, tgt=target3, init=np.array(g_init), T=np.array(g_T),
     error=np.array(g_err), corr=np.array(g_corr), iterations=np.array(g_it))
print("  G4 iterations:", g_it, "errors:", np.round(g_err, 6))

# --------------------------------------------------------------------- G5
rng = np.random.default_rng(5)
kp, kq, kT = [], [], []
for i in range(24):                      # planar; i % 3 == 0 forces a mirrored target (reflection branch)
    n = int(rng.integers(3, 40))
    p = np.column_stack((rng.normal(size=(n, 2)) * rng.uniform(0.1, 5), np.zeros(n)))
    th = rng.uniform(-np.pi, np.pi)
    q = p.copy()
    if i % 3 == 0:
        q[:, 1] *= -1
    q[:, :2] = apply_se2(se2(rng.normal(), rng.normal(), th), q[:, :2]) + rng.normal(0, 0.05, (n, 2))
    kp.append(p); kq.append(q); kT.append(ref_icp.get_transform(p, q))
for i in range(12):                      # 3-D; every 4th mirrored (det fix branch)
    n = int(rng.integers(4, 60))
    p = rng.normal(size=(n, 3)) * rng.uniform(0.1, 3, 3)
    A, _ = np.linalg.qr(rng.normal(size=(3, 3)))
    if np.linalg.det(A) < 0:
        A[:, 0] *= -1
    q = p.copy()
    if i % 4 == 0:
        q[:, 2] *= -1
    q = q @ A.T + rng.normal(size=3) + rng.normal(0, 0.02, (n, 3))
    kp.append(p); kq.append(q); kT.append(ref_icp.get_transform(p, q))
p1 = np.array([[1.0, 2.0, 0.0]]); q1 = np.array([[-0.5, 0.25, 0.0]])           # single point: R = I
kp.append(p1); kq.append(q1); kT.append(ref_icp.get_transform(p1, q1))
pc, po = ragged(kp)
qc, _ = ragged(kq)
n_refl = sum(int(T[2, 2] < 0) for T in kT[:24])
save("kabsch.npz", p=pc, q=qc, off=po, T=np.array(kT), n_planar=24, n_3d=12)
print("  G5 planar reflections:", n_refl)

# --------------------------------------------------------------------- G6
rng = np.random.default_rng(6)
N_SEQ = 32
true_poses = np.zeros((N_SEQ, 3))
for i in range(1, N_SEQ):
    T_prev = se2(*true_poses[i - 1])
    step = se2(rng.uniform(0.0, 0.05), rng.normal(0, 0.003), rng.uniform(-0.03, 0.03))
    true_poses[i] = ref_utils.pose_from_T(T_prev @ step)
odom = true_poses + np.cumsum(rng.normal(0, [0.002, 0.002, 0.001], (N_SEQ, 3)), axis=0)
odom[0] = 0
z_ts = []
for i in range(N_SEQ):
    keep = rng.random(scan.shape[0]) > 0.08
    pts = apply_se2(np.linalg.inv(se2(*true_poses[i])), scan[keep]) + rng.normal(0, 0.004, (keep.sum(), 2))
    z_ts.append(pts)
_iter_counter["n"] = 0
poses, rel = ref_loc.poses_from_scan_matching(odom, z_ts, return_relative_poses=True)
zc, zo = ragged(z_ts)
save("seq_scanmatch.npz", x_ts=odom, z=zc, z_off=zo, poses=poses, relative_poses=rel,
     total_iterations=_iter_counter["n"])
print("  G6 total iterations:", _iter_counter["n"])

# --------------------------------------------------------------------- G7
filt = ref_loc.DBSCAN_outliers_removal([scan], eps=0.1, min_samples=10)[0]
print("  filtered scan:", filt.shape)
rng = np.random.default_rng(7)
ogm_poses = np.column_stack((rng.uniform(-3, 3, 8), rng.uniform(-3, 3, 8), rng.uniform(-np.pi, np.pi, 8)))
ogm_poses[0] = 0
EXACT_STEP = 2.0 ** -10        # an exactly representable step turns log-odds into exact hit-miss counts


def ref_counts(make_grid, pose, z):
    g = make_grid()
    g.logodds_ratio = EXACT_STEP
    g.update_map(pose, z)
    c = np.rint(g.grid_map_log_odds.astype(np.float64) / EXACT_STEP)
    assert np.all(np.abs(c) < 20 / EXACT_STEP - 1)
    return c.astype(np.int32)


def coo(grid):
    ix, iy = np.nonzero(grid)
    return ix.astype(np.int32), iy.astype(np.int32), grid[ix, iy]


def ogm_fixture(name, make_grid, poses, z):
    cnt_ix, cnt_iy, cnt_v, cnt_off = [], [], [], [0]
    g = make_grid()
    for pose in poses:
        c = ref_counts(make_grid, pose, z)
        ix, iy, v = coo(c)
        cnt_ix.append(ix); cnt_iy.append(iy); cnt_v.append(v); cnt_off.append(cnt_off[-1] + ix.size)
        g.update_map(pose, z)
    lx, ly, lv = coo(g.grid_map_log_odds)
    gb = make_grid()
    gb.build_map(poses, [z] * len(poses))
    assert np.array_equal(gb.grid_map_log_odds, g.grid_map_log_odds)
    start_cells = np.array([g.world2grid(p[0] + 0.13323, p[1]) for p in poses])
    end_cells = np.array([g.world2grid(*(z @ np.array([[np.cos(p[2]), -np.sin(p[2])],
                                                       [np.sin(p[2]), np.cos(p[2])]]).T + p[:2]).T)
                          for p in poses])
    save(name, poses=poses, scan=z, cnt_ix=np.concatenate(cnt_ix), cnt_iy=np.concatenate(cnt_iy),
         cnt_v=np.concatenate(cnt_v), cnt_off=np.array(cnt_off), lo_ix=lx, lo_iy=ly, lo_v=lv,
         grid_map=np.packbits(gb.grid_map), shape=np.array(g.grid_map.shape),
         start_cells=start_cells, end_cells=end_cells,
         geometry=np.array([g.res, g.world_map_max_x, g.world_map_max_y, g.world_map_min_x,
                            g.world_map_min_y, g.buffer]))


ogm_fixture("ogm_main.npz", lambda: ref_ogm.OccupancyGridMap(0.05, 30, 30, -30, -30), ogm_poses, filt)
small_poses = ogm_poses.copy()
small_poses[3] = [5.9, -5.9, 0.7]          # origin near a corner
small_poses[4] = [7.0, 0.0, 3.0]           # ray origin outside the map
ogm_fixture("ogm_small.npz", lambda: ref_ogm.OccupancyGridMap(0.1, 6, 6, -6, -6), small_poses, scan)

# rays: reference Bresenham on random and degenerate end points
rng = np.random.default_rng(77)
g = ref_ogm.OccupancyGridMap(0.05, 30, 30, -30, -30)
rays = rng.integers(-40, 40, size=(200, 4))
rays[:8] = [[0, 1, 10, 5], [0, 1, 9, 6], [3, 3, 3, 3], [5, 5, 5, 9], [5, 5, 9, 5], [0, 0, -7, -7],
            [2, 2, -5, 3], [1, 1, 2, -30]]
cells = [g.bresenham2D(*r) for r in rays]
cc, co = ragged(cells)
save("bresenham.npz", rays=rays, cells=cc, off=co)

# --------------------------------------------------------------------- G8
rng = np.random.default_rng(8)
pts = rng.uniform(-1, 1, (1000, 3))
save("voxel.npz", points=pts, voxel_size=0.25, out=ref_icp.voxel_downsample(pts, 0.25))

# ------------------------------------------------------- reference outputs/*.npy
K = 512
out = {k: np.load(os.path.join(REF, "outputs", f"{k}_20.npy")) for k in
       ("poses_odom", "relative_poses_odom", "poses_scan_matching", "relative_poses_scan_matching")}
save("se2_outputs20.npz", poses_odom=out["poses_odom"][:K + 1],
     relative_poses_odom=out["relative_poses_odom"][:K],
     poses_scan_matching=out["poses_scan_matching"][:K + 1],
     relative_poses_scan_matching=out["relative_poses_scan_matching"][:K])
print("done")

# ------------------------------------------------- pre-processing
# get_lidar_data -> DBSCAN_outliers_removal(eps=0.1, min_samples=10) -> statistical_outliers_removal(k_std=2)
# exactly as main.py:84-87 chains them
rng = np.random.default_rng(84)
R = np.tile(ranges, (8, 1))
R[1] *= 0.6
R[2] = np.roll(ranges, 37) + rng.normal(0, 0.01, ranges.shape)
R[3, ::7] = 0.0                      # invalid beams
R[4, 100:140] = 45.0                 # beyond range_max
R[5] = ranges * rng.uniform(0.9, 1.1, ranges.shape)
R[6] = np.clip(ranges + rng.normal(0, 0.3, ranges.shape), 0.0, None)      # noisy: many DBSCAN outliers
R[7, 500:520] += 6.0                 # a far blob: radial outliers
z0 = ref_loc.get_lidar_data(R, 0.1, 30)
z1 = ref_loc.DBSCAN_outliers_removal(z0, eps=0.1, min_samples=10)
z2 = ref_loc.statistical_outliers_removal(z1, k_std=2)
a0, o0 = ragged(z0); a1, o1 = ragged(z1); a2, o2 = ragged(z2)
save("lidar_pre.npz", ranges=R, xy0=a0, off0=o0, xy1=a1, off1=o1, xy2=a2, off2=o2,
     range_min=0.1, range_max=30.0, eps=0.1, min_samples=10, k_std=2)
print("  pre-processing sizes:", np.diff(o0).tolist(), np.diff(o1).tolist(), np.diff(o2).tolist())
