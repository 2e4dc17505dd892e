"""Parity of the HIP ICP path (through the C ABI) with the oracle and the reference's golden vectors."""

import numpy as np
import pytest

from conftest import TOL_ROT, TOL_TRANS, golden, split

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["all_pairs", "auto"])
def search_default(request, monkeypatch):
    """every test of this file runs twice: with the all-pairs sweep as the library's default search strategy and
    with the automatic choice (exact pruning where supported); an explicit prune= argument wins over either"""
    if request.param == "all_pairs":
        monkeypatch.setenv("LSLAM_ICP_PRUNE", "0")
    else:
        monkeypatch.delenv("LSLAM_ICP_PRUNE", raising=False)
    return request.param


class Precision:
    """how a test hands its clouds to the library and what it may then demand:
    f32  float32 clouds: exact float32 nearest neighbours — identical to the oracle's DEVICE recipe (integer outputs
         bit-exact), relative poses within 5e-5 m / 5e-6 rad of the float64 reference;
    f64  float64 clouds: float64-exact mode — correspondences identical to the reference's float64 search (the oracle's
         f64 recipe, brute force, lowest index on exact ties) and T within 1e-9 of the reference itself."""

    def __init__(self, name):
        self.name = name
        self.dtype = np.float32 if name == "f32" else np.float64
        self.recipe = "device" if name == "f32" else "f64"
        self.tol_t, self.tol_r = (TOL_TRANS, TOL_ROT) if name == "f32" else (1e-9, 1e-9)
        self.err_rtol = 1e-6 if name == "f32" else 1e-10

    def cast(self, a):
        return np.ascontiguousarray(a, dtype=self.dtype)

    def pack(self, clouds):
        from lidar_slam_amd.batch import pack_clouds
        return pack_clouds(list(clouds), dtype=self.dtype)

    def close(self, T, T_ref, scale=1.0):
        assert np.abs(T[:3, 3] - T_ref[:3, 3]).max() < self.tol_t * scale
        assert np.abs(T[:3, :3] - T_ref[:3, :3]).max() < self.tol_r * scale
        assert np.array_equal(T[3], T_ref[3])


@pytest.fixture(autouse=True, params=["f32", "f64"])
def prec(request):
    """every test of this file also runs in both precision modes"""
    return Precision(request.param)


@pytest.fixture(scope="module")
def icp():
    import lidar_slam_amd
    assert lidar_slam_amd.device_count() >= 1, "GPU tests need a HIP device"
    from lidar_slam_amd.modules import icp as m
    return m


def close_to_reference(T, T_ref):
    """stated tolerance vs the float64 reference: 5e-5 m, 5e-6 rad (rotation entries)"""
    assert np.abs(T[:3, 3] - T_ref[:3, 3]).max() < TOL_TRANS
    assert np.abs(T[:3, :3] - T_ref[:3, :3]).max() < TOL_ROT
    assert np.array_equal(T[3], T_ref[3])


def batch_of(g, prec, key_src="src", with_init=True):
    srcs = split(g[key_src], g["src_off"])
    flat, off = prec.pack(srcs)
    S = len(srcs)
    tgt = prec.cast(np.tile(g["tgt"], (S, 1)))
    toff = np.arange(S + 1, dtype=np.int64) * g["tgt"].shape[0]
    init = g["init"] if with_init else np.tile(np.eye(4), (S, 1, 1))
    return srcs, flat, off, tgt, toff, init


def test_planar_pairs_batched_vs_golden_and_oracle(icp, prec):
    from lidar_slam_amd.batch import icp_batch
    from oracle import icp_oracle
    g = golden("icp_planar.npz")
    srcs, flat, off, tgt, toff, init = batch_of(g, prec)
    r = icp_batch(flat, off, tgt, toff, init, epsilon=float(g["epsilon"]), return_correspondences=True)
    np.testing.assert_array_equal(r.iterations, g["iterations"])          # same number of iterations as the reference
    np.testing.assert_array_equal(r.correspondences, g["corr"])           # the REFERENCE's correspondences, every one
    for k, src in enumerate(srcs):
        prec.close(r.T[k], g["T"][k])
        assert abs(r.error[k] - g["error"][k]) < prec.err_rtol * max(1.0, g["error"][k])
        o = icp_oracle.run_icp_full(src, g["tgt"], init[k], epsilon=float(g["epsilon"]), recipe=prec.recipe, nn="brute")
        np.testing.assert_array_equal(r.correspondences[off[k]:off[k + 1]], o.correspondences)   # bit-exact indices
        np.testing.assert_allclose(r.T[k], o.T, atol=1e-10)
        np.testing.assert_allclose(r.error[k], o.error, rtol=1e-10)


def test_run_icp_call_surface(icp, prec):
    g = golden("icp_planar.npz")
    src = g["src"][: g["src_off"][1]]
    src_before = src.copy()
    T = icp.run_icp(src, g["tgt"], init_transform=g["init"][0], epsilon=0.001)
    close_to_reference = Precision("f64").close                # float64 arrays in: float64-exact mode, whatever `prec` is
    assert isinstance(T, np.ndarray) and T.shape == (4, 4) and T.dtype == np.float64
    T2, e = icp.run_icp(src, g["tgt"], g["init"][0], 0.001, return_error=True)
    T3, c = icp.run_icp(src, g["tgt"], g["init"][0], 0.001, return_correspondences=True)
    T4, e4, c4 = icp.run_icp(src, g["tgt"], g["init"][0], 0.001, return_error=True, return_correspondences=True)
    assert np.array_equal(T, T2) and np.array_equal(T, T3) and np.array_equal(T, T4)
    assert e == e4 and isinstance(e, np.float64) and c.dtype == np.int64 and np.array_equal(c, c4)
    assert np.array_equal(src, src_before)                     # inputs never mutated (I8)
    close_to_reference(T, g["T"][0])
    # float32 input is accepted (promoted in the reference; searched as float32 here)
    Precision("f32").close(icp.run_icp(src.astype(np.float32), g["tgt"].astype(np.float32), g["init"][0], 0.001), g["T"][0])


def test_forced_50_iterations(icp, prec):
    from lidar_slam_amd.batch import icp_batch
    g = golden("icp_forced50.npz")
    srcs, flat, off, tgt, toff, init = batch_of(g, prec)
    r = icp_batch(flat, off, tgt, toff, init, epsilon=0.0, stopping_thresh=0.0, max_iters=49)
    assert r.iterations.tolist() == [50, 50]
    for k in range(2):
        prec.close(r.T[k], g["T"][k])


def test_loop_closure_style_normalized_error(icp, prec):
    g = golden("icp_loopclose.npz")
    srcs, corrs = split(g["src"], g["src_off"]), split(g["corr"], g["src_off"])
    for k, src in enumerate(srcs):
        T, err, corr = icp.run_icp(prec.cast(src), prec.cast(g["tgt"]), epsilon=float(g["epsilon"][k]), return_error=True,
                                   normalize_error=True, return_correspondences=True)
        prec.close(T, g["T"][k])
        assert abs(err - g["error"][k]) < prec.err_rtol * g["error"][k] + 1e-12
        np.testing.assert_array_equal(corr, corrs[k])             # the reference's correspondences, every one


def test_3d_multistart_vs_golden_and_oracle(icp, prec):
    from lidar_slam_amd.batch import icp_batch
    from oracle import c_oracle
    g = golden("icp_3d.npz")
    S = g["init"].shape[0]
    n, m = g["src"].shape[0], g["tgt"].shape[0]
    src = prec.cast(np.tile(g["src"], (S, 1)))
    tgt = prec.cast(np.tile(g["tgt"], (S, 1)))
    r = icp_batch(src, np.arange(S + 1) * n, tgt, np.arange(S + 1) * m, g["init"], epsilon=0.001,
                  normalize_error=True, return_correspondences=True)
    np.testing.assert_array_equal(r.iterations, g["iterations"])
    np.testing.assert_array_equal(r.correspondences, g["corr"].reshape(-1))      # the reference's correspondences, every one
    for k in range(S):
        prec.close(r.T[k], g["T"][k])
        To, eo, co, ito = c_oracle.run_icp(g["src"], g["tgt"], g["init"][k], epsilon=0.001, normalize_error=True,
                                           recipe=prec.recipe)
        np.testing.assert_array_equal(r.correspondences[k * n:(k + 1) * n], co)
        np.testing.assert_allclose(r.T[k], To, atol=1e-9)
        np.testing.assert_allclose(r.error[k], eo, rtol=1e-9)


@pytest.mark.parametrize("sequential", [False, True])
def test_register_multi_start_matches_reference_loop(icp, sequential, monkeypatch, prec):
    """24 yaw starts of code/icp_warm_up/test_icp.py:66-92: one batched launch, or whole-chip runs back to back.
    float64 clouds (float64-exact mode): EVERY start equals the reference's, within 1e-9 — no exception.  float32 clouds:
    rounding the inputs flips a correspondence in one wrong-basin start (start 15, 1.1e-4 rad off); that start must
    then equal the float64 algorithm run on the ROUNDED clouds — the documented limit of handing over float32 points."""
    from lidar_slam_amd import icp_warm_up as W
    g, m = golden("icp_3d.npz"), golden("icp_multistart.npz")
    if sequential:
        monkeypatch.setattr(W, "SEQUENTIAL_PAIR_EVALS", 0)
    from oracle import icp_oracle
    pose, err, poses, errors, iters = W.register_multi_start(prec.cast(g["src"]), prec.cast(g["tgt"]), return_all=True)
    np.testing.assert_array_equal(iters, m["iterations"])
    init = W.yaw_starts(g["src"], g["tgt"])
    r32 = lambda a: a.astype(np.float32).astype(np.float64)
    sensitive = 0
    for k in range(24):
        if np.abs(poses[k][:3, :3] - m["T"][k][:3, :3]).max() < prec.tol_r:
            prec.close(poses[k], m["T"][k])
            assert abs(errors[k] - m["error"][k]) < 20 * prec.err_rtol * m["error"][k]
            continue
        sensitive += 1
        o = icp_oracle.run_icp_full(r32(g["src"]), r32(g["tgt"]), init[k], epsilon=0.001, normalize_error=True)
        assert np.abs(o.T[:3, :3] - m["T"][k][:3, :3]).max() > TOL_ROT
        prec.close(poses[k], o.T)
    assert sensitive == (0 if prec.name == "f64" else 1)
    assert int(np.argmin(errors)) == int(m["best"])
    prec.close(pose, m["best_pose"])
    assert isinstance(err, np.float64) and abs(err - m["best_error"]) < 20 * prec.err_rtol * m["best_error"]
    np.testing.assert_allclose(W.yaw_starts(g["src"], g["tgt"])[[0, 4, 5, 13]], g["init"], atol=1e-12)


def test_register_multi_start_many_equals_one_target_at_a_time(icp):
    from lidar_slam_amd import icp_warm_up as W
    g = golden("icp_3d.npz")
    rng = np.random.default_rng(4)
    targets = [g["tgt"], g["tgt"][::2] + 0.001, g["tgt"][100:] @ np.array([[0.0, -1, 0], [1, 0, 0], [0, 0, 1.0]])]
    many = W.register_multi_start_many(g["src"], targets)
    for (pose, err), t in zip(many, targets):
        p1, e1 = W.register_multi_start(g["src"], t)
        np.testing.assert_array_equal(pose, p1)
        assert err == e1
    assert W.register_multi_start_many(g["src"], []) == []


def test_clouds_far_from_the_origin_are_centred_before_rounding(icp, prec):
    """at |coordinate| = 1e5 m a float32 holds 8 mm; run_icp shifts such clouds in float64 first"""
    from oracle import icp_oracle
    rng = np.random.default_rng(3)
    tgt = rng.uniform(-5, 5, (300, 2)) + np.array([1.0e5, -2.0e5])
    th = 0.02
    R = np.array([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]])
    c = tgt.mean(0)
    src = (tgt[:250] - c) @ R.T + c + np.array([0.03, -0.02]) + rng.normal(0, 0.002, (250, 2))
    if prec.name == "f32":
        pytest.skip("float32 arrays cannot hold these coordinates; run_icp shifts float64 input before rounding")
    T, err, corr = icp.run_icp(src, tgt, return_error=True, return_correspondences=True, epsilon=1e-9)
    o = icp_oracle.run_icp_full(src, tgt, None, epsilon=1e-9)
    assert np.array_equal(corr, o.correspondences)
    assert abs(err - o.error) < 1e-5 * o.error                 # float32 rounding of the centred points (±5 m)
    p = np.c_[src, np.zeros(len(src)), np.ones(len(src))]
    assert np.abs(p @ T.T - p @ o.T.T).max() < 1e-4            # same alignment of the points (t alone is R·1e5-conditioned)
    assert np.abs(T[:3, :3] - o.T[:3, :3]).max() < TOL_ROT


def test_get_correspondences_in_both_precisions(icp, prec):
    """modules/icp.py:29-41: float64 points -> the float64 search's indices (duplicates: lowest index; near ties decided in
    float64), float32 points -> the exact float32 recipe; 3-D and planar-padded, a target at the rim of float32 resolution"""
    from oracle import c_oracle, icp_oracle
    rng = np.random.default_rng(17)
    tgt = rng.uniform(-30, 30, (6000, 3))
    src = tgt[rng.integers(0, 6000, 2500)] + rng.normal(0, 0.02, (2500, 3))
    tgt[4000:4010] = tgt[20:30]                                   # exact duplicates: the lowest index must win
    # two targets whose distances to a source differ by less than float32 can tell: in float32 an exact tie (lowest index, 100),
    # in float64 the second one is nearer by 3 nm
    src[5] = [100.0, 100.0, 100.0]
    tgt[100] = src[5] + np.array([0.25 + 3e-9, 0.0, 0.0]); tgt[101] = src[5] - np.array([0.25, 0.0, 0.0])
    s, t = prec.cast(src), prec.cast(tgt)
    idx = icp.get_correspondences(s, icp.TargetCloud(t))
    assert idx.dtype == np.int64
    if prec.name == "f32":
        ref, _ = c_oracle.nn_f32(s, t)
        assert idx[5] == 100
    else:
        ref = icp_oracle.nn_bruteforce(src, tgt)
        assert idx[5] == 101
    np.testing.assert_array_equal(idx, ref)


def test_get_transform_get_error_get_correspondences(icp, prec):
    if prec.name == "f64":
        pytest.skip("these operators have one precision")
    from oracle import c_oracle
    g = golden("kabsch.npz")
    for k, (p, q) in enumerate(zip(split(g["p"], g["off"]), split(g["q"], g["off"]))):
        np.testing.assert_allclose(icp.get_transform(p, q), g["T"][k], atol=1e-10)
    rng = np.random.default_rng(1)
    p, q = rng.normal(size=(777, 3)), rng.normal(size=(777, 3))
    assert abs(icp.get_error(p, q) - np.sum((p - q) ** 2)) < 1e-9
    both = np.vstack((p, q)); diag2 = np.sum((both.max(0) - both.min(0)) ** 2)
    assert abs(icp.get_error(p, q, normalize=True) - np.sum((p - q) ** 2) / (diag2 * 777)) < 1e-12
    tgt = rng.uniform(-5, 5, (5000, 3))
    idx = icp.get_correspondences(p, icp.TargetCloud(tgt))
    ref, _ = c_oracle.nn_f32(p.astype(np.float32), tgt.astype(np.float32))
    assert idx.dtype == np.int64 and np.array_equal(idx, ref)


def test_icp_iteration_is_one_step(icp, prec):
    from oracle import icp_oracle
    g = golden("icp_planar.npz")
    src = prec.cast(icp_oracle.as_xyz(g["src"][: g["src_off"][1]]))
    tgt = prec.cast(icp_oracle.as_xyz(g["tgt"]))
    T, corr, err = icp.icp_iteration(np.hstack((src, np.ones((src.shape[0], 1), src.dtype))),
                                     np.hstack((tgt, np.ones((tgt.shape[0], 1), tgt.dtype))), icp.TargetCloud(tgt), g["init"][0])
    To, co, eo = icp_oracle.icp_iteration(src.astype(np.float64), tgt.astype(np.float64), g["init"][0], recipe=prec.recipe, planar=False)
    np.testing.assert_array_equal(corr, co)
    np.testing.assert_allclose(T, To, atol=1e-10)
    np.testing.assert_allclose(err, eo, rtol=1e-10)


def test_errors_match_the_reference(icp, prec):
    g = golden("icp_planar.npz")
    src = prec.cast(g["src"][:100])
    with pytest.raises(IndexError):
        icp.run_icp(src, np.zeros((0, 2), prec.dtype))
    bad = src.copy(); bad[7, 1] = np.inf
    with pytest.raises(ValueError):
        icp.run_icp(bad, prec.cast(g["tgt"]))
    T, err, corr = icp.run_icp(np.zeros((0, 2), prec.dtype), prec.cast(g["tgt"]), return_error=True, return_correspondences=True)
    assert np.all(np.isnan(T[:3])) and np.array_equal(T[3], [0, 0, 0, 1]) and err == 0.0 and corr.size == 0


def test_scan_matching_sequence_vs_golden(icp, prec):
    from lidar_slam_amd.modules.localization import poses_from_scan_matching, scan_matching_relative_poses
    g = golden("seq_scanmatch.npz")
    z = [prec.cast(a) for a in split(g["z"], g["z_off"])]
    poses, rel = poses_from_scan_matching(g["x_ts"], z, return_relative_poses=True)
    assert poses.shape == g["poses"].shape and rel.shape == g["relative_poses"].shape
    assert np.abs(rel[:, :2, 2] - g["relative_poses"][:, :2, 2]).max() < prec.tol_t
    assert np.abs(rel[:, :2, :2] - g["relative_poses"][:, :2, :2]).max() < prec.tol_r
    assert np.abs(poses[:, :2] - g["poses"][:, :2]).max() < 31 * prec.tol_t
    _, info = scan_matching_relative_poses(g["x_ts"], z, return_info=True)
    assert int(info.iterations.sum()) == int(g["total_iterations"])
    assert np.array_equal(poses_from_scan_matching(g["x_ts"], z), poses)


@pytest.mark.parametrize("env", [{"LSLAM_ICP_POINTS_PER_LANE": "1"}, {"LSLAM_ICP_POINTS_PER_LANE": "4"},
                                 {"LSLAM_ICP_TILE_POINTS": "256"}, {"LSLAM_ICP_THREADS": "128"},
                                 {"LSLAM_ICP_TILE_POINTS": "128", "LSLAM_ICP_THREADS": "64", "LSLAM_ICP_POINTS_PER_LANE": "2"}])
def test_launch_geometry_does_not_change_results(icp, env, monkeypatch, prec):
    """register blocking, multi-pass sources and multi-tile targets all give the same answers"""
    from lidar_slam_amd.batch import icp_batch
    g = golden("icp_planar.npz")
    srcs, flat, off, tgt, toff, init = batch_of(g, prec)
    base = icp_batch(flat, off, tgt, toff, init, epsilon=0.001, return_correspondences=True)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    r = icp_batch(flat, off, tgt, toff, init, epsilon=0.001, return_correspondences=True)
    np.testing.assert_array_equal(r.iterations, base.iterations)
    np.testing.assert_array_equal(r.correspondences, base.correspondences)
    np.testing.assert_allclose(r.T, base.T, atol=1e-11)


@pytest.mark.parametrize("coop", ["1", "0"])
def test_large_path_equals_batch_path(icp, prec, coop, monkeypatch):
    """the whole-chip paths (the co-resident one-launch kernel; one fused or three launches per iteration) and the
    one-workgroup path agree"""
    from lidar_slam_amd import _lib
    from lidar_slam_amd._lib import check, lib, ptr
    from lidar_slam_amd.batch import icp_batch
    monkeypatch.setenv("LSLAM_ICP_COOP", coop)
    g = golden("icp_planar.npz")
    src = prec.cast(g["src"][: g["src_off"][1]])
    tgt = prec.cast(g["tgt"])
    entry = lib().lslam_icp_large_f64 if prec.name == "f64" else lib().lslam_icp_large
    for dim, norm in ((2, False), (3, True)):
        s = src if dim == 2 else np.ascontiguousarray(np.hstack((src, np.zeros((src.shape[0], 1), src.dtype))))
        t = tgt if dim == 2 else np.ascontiguousarray(np.hstack((tgt, np.zeros((tgt.shape[0], 1), tgt.dtype))))
        T = np.empty((4, 4)); err = np.empty(1); it = np.empty(1, np.int32); corr = np.empty(s.shape[0], np.int32)
        T0 = np.ascontiguousarray(g["init"][0])
        check(entry(ptr(s), s.shape[0], ptr(t), t.shape[0], dim, ptr(T0), 1e-3 if not norm else 1e-7,
                    1e-4, 2000, _lib.ICP_NORMALIZE_ERROR if norm else 0, ptr(T), ptr(err), ptr(it), ptr(corr), 0))
        r = icp_batch(s, [0, s.shape[0]], t, [0, t.shape[0]], T0[None], epsilon=1e-3 if not norm else 1e-7,
                      normalize_error=norm, return_correspondences=True)
        assert it[0] == r.iterations[0]
        np.testing.assert_array_equal(corr, r.correspondences)
        np.testing.assert_allclose(T, r.T[0], atol=1e-10)
        np.testing.assert_allclose(err[0], r.error[0], rtol=1e-9)


def test_full_size_batch_properties(icp, prec):
    """BASELINE config sizes (4096 × 4096 planar), checked through size-independent properties:
    a known rigid motion is recovered; results do not depend on which other problems share the launch;
    permuting the target permutes the correspondences and leaves the pose alone."""
    from lidar_slam_amd.batch import icp_batch
    from lidar_slam_amd.synth import batch_c4
    S = 48
    w = batch_c4(S, 4096, seed=123)
    w.src, w.tgt = prec.cast(w.src), prec.cast(w.tgt)
    r = icp_batch(w.src, w.src_off, w.tgt, w.tgt_off, w.init_T, epsilon=1e-3, return_correspondences=True)
    err_t = np.abs(r.T[:, :2, 3] - w.true_T[:, :2, 3]).max()
    err_R = np.abs(r.T[:, :2, :2] - w.true_T[:, :2, :2]).max()
    assert err_t < 2e-3 and err_R < 2e-4, (err_t, err_R)              # 1 cm noise on 4096 points
    sub = slice(5 * 4096, 6 * 4096)
    one = icp_batch(w.src[sub], [0, 4096], w.tgt[sub], [0, 4096], w.init_T[5:6], epsilon=1e-3, return_correspondences=True)
    assert one.iterations[0] == r.iterations[5]
    np.testing.assert_array_equal(one.correspondences, r.correspondences[sub])
    np.testing.assert_array_equal(one.T[0], r.T[5])                   # bitwise: fixed-order reductions
    perm = np.random.default_rng(0).permutation(4096)
    pt = icp_batch(w.src[sub], [0, 4096], np.ascontiguousarray(w.tgt[sub][perm]), [0, 4096], w.init_T[5:6],
                   epsilon=1e-3, return_correspondences=True)
    assert pt.iterations[0] == one.iterations[0]
    np.testing.assert_array_equal(perm[pt.correspondences], one.correspondences)
    np.testing.assert_allclose(pt.T[0], one.T[0], atol=1e-12)


def test_dense_3d_nn_against_exact_c_oracle(icp, prec):
    if prec.name == "f64":
        pytest.skip("lslam_nn is the float32 search")
    """100k-style 3-D search (here 20k × 50k so that the CPU oracle finishes in seconds): every index exact"""
    import torch
    from lidar_slam_amd.batch import nn_dev
    from oracle import c_oracle
    rng = np.random.default_rng(5)
    src = rng.normal(size=(20000, 3)).astype(np.float32)
    tgt = rng.normal(size=(50000, 3)).astype(np.float32)
    tgt[1000:1010] = tgt[10:20]                       # exact duplicates: lowest index must win
    d_s, d_t = torch.from_numpy(src).cuda(), torch.from_numpy(tgt).cuda()
    idx = torch.empty(src.shape[0], dtype=torch.int32, device="cuda")
    d2 = torch.empty(src.shape[0], dtype=torch.float32, device="cuda")
    nn_dev(d_s.data_ptr(), src.shape[0], d_t.data_ptr(), tgt.shape[0], 3, None, idx.data_ptr(), d2.data_ptr(),
           stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    ref_idx, ref_d2 = c_oracle.nn_f32(src, tgt)
    np.testing.assert_array_equal(idx.cpu().numpy(), ref_idx)
    np.testing.assert_array_equal(d2.cpu().numpy(), ref_d2)


def _random_problem(rng, dim, n, m, spread):
    tgt = rng.normal(size=(m, dim)) * spread
    th = rng.uniform(-0.2, 0.2)
    R = np.eye(dim)
    R[:2, :2] = [[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]]
    pick = rng.integers(0, m, n)
    src = (tgt[pick] - rng.normal(0, 0.05 * spread, dim)) @ R + rng.normal(0, 0.01 * spread, (n, dim))
    return src, tgt


@pytest.mark.parametrize("dim", [2, 3])
def test_ragged_random_batch_matches_exact_oracle(icp, dim, prec):
    """ragged sizes from 5 points to a few thousand (rank-deficient toy cases are parity-undefined in the reference), duplicates included: identical iteration
    counts and correspondences as the C oracle's device recipe (exact fmaf), T to 1e-9"""
    from lidar_slam_amd.batch import icp_batch, pack_clouds
    from oracle import c_oracle
    rng = np.random.default_rng(100 + dim)
    sizes = [(8, 6), (5, 1500), (17, 16), (64, 65), (65, 64), (200, 33), (1000, 999), (1081, 1081), (2500, 700), (513, 4100)]
    srcs, tgts = [], []
    for n, m in sizes:
        s, t = _random_problem(rng, dim, n, m, spread=rng.uniform(0.5, 20.0))
        if m > 40:
            t[m // 2] = t[3]                          # an exact duplicate target: lowest index must win
        srcs.append(s); tgts.append(t)
    fs, so = prec.pack(srcs)
    ft, to = prec.pack(tgts)
    init = np.tile(np.eye(4), (len(sizes), 1, 1))
    r = icp_batch(fs, so, ft, to, init, epsilon=1e-12, max_iters=25, return_correspondences=True)
    for k, (s, t) in enumerate(zip(srcs, tgts)):
        To, eo, co, ito = c_oracle.run_icp(s, t, np.eye(4), epsilon=1e-12, max_iters=25, recipe=prec.recipe)
        assert r.iterations[k] == ito, (k, sizes[k])
        np.testing.assert_array_equal(r.correspondences[so[k]:so[k + 1]], co, err_msg=str(sizes[k]))
        np.testing.assert_allclose(r.T[k], To, atol=1e-9, err_msg=str(sizes[k]))
        np.testing.assert_allclose(r.error[k], eo, rtol=1e-9, atol=1e-18)


@pytest.mark.parametrize("dim,n,m,fuse", [(2, 700, 5000, 1), (2, 700, 5000, 0), (3, 9000, 3000, 1), (2, 70000, 1200, 1),
                                           (3, 300, 40000, 1), (3, 300, 40000, 0), (2, 8191, 900, 1)])
def test_whole_chip_path_random_shapes_match_exact_oracle(icp, dim, n, m, fuse, monkeypatch, prec):
    """the whole-chip path over its geometry regimes (P = 1/2/4, one and many splits; below 8192 source points the
    co-resident one-launch kernel where the target fits its LDS, else one fused launch per iteration, or three with
    LSLAM_ICP_FUSE_TAIL=0 — which also switches the co-resident kernel off)"""
    from lidar_slam_amd import _lib
    monkeypatch.setenv("LSLAM_ICP_FUSE_TAIL", str(fuse))
    if not fuse:
        monkeypatch.setenv("LSLAM_ICP_COOP", "0")
    from lidar_slam_amd._lib import check, lib, ptr
    from oracle import c_oracle
    rng = np.random.default_rng(n + m)
    s, t = _random_problem(rng, dim, n, m, spread=3.0)
    sc, tc = prec.cast(s), prec.cast(t)
    T = np.empty((4, 4)); err = np.empty(1); it = np.empty(1, np.int32); corr = np.empty(n, np.int32)
    T0 = np.eye(4)
    entry = lib().lslam_icp_large_f64 if prec.name == "f64" else lib().lslam_icp_large
    check(entry(ptr(sc), n, ptr(tc), m, dim, ptr(T0), 1e-12, 1e-4, 6, 0, ptr(T), ptr(err), ptr(it), ptr(corr), 0))
    To, eo, co, ito = c_oracle.run_icp(s, t, np.eye(4), epsilon=1e-12, max_iters=6, recipe=prec.recipe)
    assert it[0] == ito
    np.testing.assert_array_equal(corr, co)
    np.testing.assert_allclose(T, To, atol=1e-9)
    np.testing.assert_allclose(err[0], eo, rtol=1e-9)
    if fuse:                                                   # the pruned whole-chip search (box hierarchy; in float64-exact mode with its refine launch)
        check(entry(ptr(sc), n, ptr(tc), m, dim, ptr(T0), 1e-12, 1e-4, 6, _lib.ICP_PRUNE, ptr(T), ptr(err), ptr(it), ptr(corr), 0))
        assert it[0] == ito
        np.testing.assert_array_equal(corr, co)
        np.testing.assert_allclose(T, To, atol=1e-9)


def test_loop_closure_batch_matches_per_pair_reference_semantics(icp, tmp_path, prec):
    """main.py:158-172 as one launch: source = scan i, target = scan i+K, identity init,
    normalize_error=True, default epsilon; gating and file outputs keep the reference's shapes"""
    from lidar_slam_amd.modules.localization import loop_closure_constraints, save_scan_matching_outputs
    from oracle import icp_oracle
    g = golden("seq_scanmatch.npz")
    z = [prec.cast(a) for a in split(g["z"], g["z_off"])]
    pairs, T3, errors, ok = loop_closure_constraints(z, fixed_interval=10, max_distance=0.25, max_yaw=2.0)
    assert pairs.tolist() == [[0, 10], [10, 20], [20, 30]]
    for (i, j), T, e in zip(pairs, T3, errors):
        o = icp_oracle.run_icp_full(split(g["z"], g["z_off"])[i], split(g["z"], g["z_off"])[j], None, normalize_error=True)   # float64 reference semantics
        assert np.abs(T[:2, 2] - o.T[:2, 3]).max() < prec.tol_t and np.abs(T[:2, :2] - o.T[:2, :2]).max() < prec.tol_r
        assert abs(e - o.error) < prec.err_rtol * o.error + 1e-12
        ang = np.rad2deg(np.arctan2(T[1, 0], T[0, 0]))
        assert ok[list(pairs[:, 0]).index(i)] == (np.linalg.norm(T[:2, 2]) < 0.25 and ang < 2.0)
    paths = save_scan_matching_outputs(g["poses"], g["relative_poses"], 20, out_dir=str(tmp_path))
    assert np.load(paths[0]).shape == (32, 3) and np.load(paths[1]).shape == (31, 3, 3)


@pytest.mark.parametrize("plist", ["1", "0"])
def test_exact_pruning_gives_the_brute_force_answers(icp, prec, plist, monkeypatch):
    """LSLAM_ICP_PRUNE: Morton-sorted clouds + box tests must not change a single correspondence or
    iteration count (duplicates, ragged sizes, several passes per problem, normalised error) — with the chunks evaluated
    by the whole wave (plist 0) and from per-lane lists (plist 1; the launcher picks by size and precision mode)"""
    monkeypatch.setenv("LSLAM_ICP_PRUNE_LIST", plist)
    from lidar_slam_amd.batch import icp_batch
    from lidar_slam_amd.synth import batch_c4
    pack_clouds = prec.pack
    g = golden("icp_planar.npz")
    srcs, flat, off, tgt, toff, init = batch_of(g, prec)
    a = icp_batch(flat, off, tgt, toff, init, epsilon=0.001, return_correspondences=True, prune=False)
    b = icp_batch(flat, off, tgt, toff, init, epsilon=0.001, return_correspondences=True, prune=True)
    np.testing.assert_array_equal(b.iterations, a.iterations)
    np.testing.assert_array_equal(b.correspondences, a.correspondences)
    np.testing.assert_allclose(b.T, a.T, atol=1e-11)
    np.testing.assert_allclose(b.error, a.error, rtol=1e-11)

    rng = np.random.default_rng(77)
    sizes = [(8, 6), (100, 3000), (4096, 4096), (3000, 700), (1081, 1081), (4000, 64), (513, 4096)]
    ss, ts = [], []
    for n, m in sizes:
        s_, t_ = _random_problem(rng, 2, n, m, spread=rng.uniform(0.5, 20.0))
        if m > 40:
            t_[m // 2] = t_[3]; t_[m - 1] = t_[3]          # exact duplicates far apart in index
        ss.append(s_); ts.append(t_)
    fs, so = pack_clouds(ss)
    ft, to = pack_clouds(ts)
    I = np.tile(np.eye(4), (len(sizes), 1, 1))
    for norm in (False, True):
        a = icp_batch(fs, so, ft, to, I, epsilon=1e-12, max_iters=12, normalize_error=norm, return_correspondences=True, prune=False)
        b = icp_batch(fs, so, ft, to, I, epsilon=1e-12, max_iters=12, normalize_error=norm, return_correspondences=True, prune=True)
        np.testing.assert_array_equal(b.iterations, a.iterations)
        np.testing.assert_array_equal(b.correspondences, a.correspondences)
        np.testing.assert_allclose(b.T, a.T, atol=1e-10)

    w = batch_c4(24, 4096, seed=5)                           # the benchmark's shape
    w.src, w.tgt = prec.cast(w.src), prec.cast(w.tgt)
    a = icp_batch(w.src, w.src_off, w.tgt, w.tgt_off, w.init_T, epsilon=0.0, stopping_thresh=0.0, max_iters=9, return_correspondences=True, prune=False)
    b = icp_batch(w.src, w.src_off, w.tgt, w.tgt_off, w.init_T, epsilon=0.0, stopping_thresh=0.0, max_iters=9, return_correspondences=True, prune=True)
    np.testing.assert_array_equal(b.correspondences, a.correspondences)
    np.testing.assert_allclose(b.T, a.T, atol=1e-11)

    # above the size limit the flag is ignored (plain sweep), not an error
    s_, t_ = _random_problem(rng, 2, 5000, 5000, 3.0)
    a = icp_batch(*pack_clouds([s_]), *pack_clouds([t_]), np.eye(4)[None], max_iters=3, epsilon=1e-12, return_correspondences=True, prune=False)
    b = icp_batch(*pack_clouds([s_]), *pack_clouds([t_]), np.eye(4)[None], max_iters=3, epsilon=1e-12, return_correspondences=True, prune=True)
    np.testing.assert_array_equal(b.correspondences, a.correspondences)
    np.testing.assert_array_equal(b.T, a.T)


@pytest.mark.parametrize("keep", ["auto", "0", "1", "2", "3"])
def test_pruned_search_is_exact_for_any_point_order(icp, prec, keep, monkeypatch):
    """the pruned batch path keeps a cloud in the order it arrives when that order is compact (LiDAR beam order) and
    Morton-sorts it otherwise: scans in beam order, the same scans shuffled, one of each, collinear and single-point clouds —
    the decision (auto) and every forced combination (bit 0: target as given, bit 1: source as given) must give the all-pairs
    sweep's correspondences and iteration counts"""
    from lidar_slam_amd.batch import icp_batch
    from lidar_slam_amd.synth import sequence_c3
    from lidar_slam_amd.modules.localization import relative_pose_guesses
    pack_clouds = prec.pack
    if keep != "auto":
        monkeypatch.setenv("LSLAM_ICP_PRUNE_KEEP", keep)
    rng = np.random.default_rng(2024)
    seq = sequence_c3(9, seed=11)
    scans = [np.asarray(z, dtype=np.float64) for z in seq.z_ts]
    init = list(relative_pose_guesses(seq.x_ts))                      # pair i: scan i+1 onto scan i
    srcs, tgts = [z for z in scans[1:]], [z for z in scans[:-1]]
    for k in (1, 4):                                                   # source shuffled
        srcs[k] = srcs[k][rng.permutation(srcs[k].shape[0])]
    for k in (2, 4):                                                   # target shuffled
        tgts[k] = tgts[k][rng.permutation(tgts[k].shape[0])]
    fs, so = pack_clouds(srcs)
    ft, to = pack_clouds(tgts)
    I = np.stack(init)
    kw = dict(epsilon=1e-3, return_correspondences=True)
    a = icp_batch(fs, so, ft, to, I, prune=False, **kw)
    b = icp_batch(fs, so, ft, to, I, prune=True, **kw)
    np.testing.assert_array_equal(b.iterations, a.iterations)
    np.testing.assert_array_equal(b.correspondences, a.correspondences)
    np.testing.assert_allclose(b.T, a.T, atol=1e-10)
    # clouds on a line (ordered and shuffled) and clouds collapsed to a point: the order test must cope with zero extents; their
    # Kabsch step is rank-deficient (parity undefined), so one search from the identity is compared, not the transforms
    line = np.column_stack((np.linspace(-3, 5, 700), np.full(700, 0.25)))
    srcs = [line[:650] + [0.01, 0.0], line[rng.permutation(700)], np.array([[0.5, 0.5]] * 80), line[::-1]]
    tgts = [line, line + [0.0, 0.02], np.array([[0.6, 0.4]] * 70), line[rng.permutation(700)]]
    fs, so = pack_clouds(srcs)
    ft, to = pack_clouds(tgts)
    I = np.tile(np.eye(4), (len(srcs), 1, 1))
    kw = dict(epsilon=0.0, stopping_thresh=0.0, max_iters=0, return_correspondences=True)
    a = icp_batch(fs, so, ft, to, I, prune=False, **kw)
    b = icp_batch(fs, so, ft, to, I, prune=True, **kw)
    np.testing.assert_array_equal(b.correspondences, a.correspondences)
    np.testing.assert_allclose(b.error, a.error, rtol=1e-12)


@pytest.mark.parametrize("plist", ["1", "0"])
def test_pruning_on_degenerate_clouds_with_massive_ties(icp, prec, plist, monkeypatch):
    """lattices, coincident points, a cloud collapsed to one point: every tie must still resolve to the
    lowest ORIGINAL index, exactly as the all-pairs sweep does (both ways of evaluating the surviving chunks)"""
    monkeypatch.setenv("LSLAM_ICP_PRUNE_LIST", plist)
    from lidar_slam_amd.batch import icp_batch
    pack_clouds = prec.pack
    rng = np.random.default_rng(5)
    gx, gy = np.meshgrid(np.arange(40) * 0.25, np.arange(40) * 0.25)
    lattice = np.column_stack((gx.ravel(), gy.ravel()))                      # 1600 points, exact ties everywhere
    shuffled = lattice[rng.permutation(lattice.shape[0])]
    cases = [
        (lattice + 0.125, shuffled),                                          # sources at cell centres: 4-way ties
        (shuffled[:700], np.vstack((lattice, lattice[::-1]))),                # every target duplicated
        (rng.normal(size=(500, 2)), np.tile([[1.5, -2.0]], (300, 1))),        # target collapsed to one point
        (np.tile([[0.3, 0.3]], (200, 1)), lattice),                           # source collapsed to one point
        (lattice[:64], lattice[:64]),                                         # identical clouds
    ]
    fs, so = pack_clouds([c[0] for c in cases])
    ft, to = pack_clouds([c[1] for c in cases])
    I = np.tile(np.eye(4), (len(cases), 1, 1))
    a = icp_batch(fs, so, ft, to, I, epsilon=1e-30, max_iters=2, return_correspondences=True, prune=False)
    b = icp_batch(fs, so, ft, to, I, epsilon=1e-30, max_iters=2, return_correspondences=True, prune=True)
    np.testing.assert_array_equal(b.correspondences, a.correspondences)
    np.testing.assert_array_equal(b.iterations, a.iterations)
    # cases 2 and 3 (a cloud collapsed to one point) have a rank-deficient cross-covariance: rotation undefined
    defined = [0, 1, 4]
    assert np.isfinite(a.T[defined]).all() and np.isfinite(b.T[defined]).all()
    np.testing.assert_allclose(b.T[defined], a.T[defined], atol=1e-9)
    # first iteration of case 0 against a direct evaluation: lowest index among the 4 equidistant corners
    s32, t32 = prec.cast(lattice + 0.125), prec.cast(shuffled)
    one = icp_batch(s32, [0, 1600], t32, [0, 1600], np.eye(4)[None], epsilon=0.0, max_iters=0, stopping_thresh=0.0,
                    return_correspondences=True, prune=True)
    d = ((s32[:, None, :] - t32[None, :, :]) ** 2).sum(-1)
    np.testing.assert_array_equal(one.correspondences, np.argmin(d, axis=1))


def test_module_switch_for_pruning_keeps_run_icp_results(icp, monkeypatch, prec):
    g = {k: (prec.cast(v) if k in ("src", "tgt") else v) for k, v in golden("icp_3d.npz").items()}
    close_to_reference = prec.close
    monkeypatch.setattr(icp, "PRUNE", False)
    ref = icp.run_icp(g["src"], g["tgt"], g["init"][1], 0.001, return_error=True, normalize_error=True, return_correspondences=True)
    monkeypatch.setattr(icp, "PRUNE", True)
    got = icp.run_icp(g["src"], g["tgt"], g["init"][1], 0.001, return_error=True, normalize_error=True, return_correspondences=True)
    np.testing.assert_array_equal(got[2], ref[2])
    np.testing.assert_allclose(got[0], ref[0], atol=1e-11)
    np.testing.assert_allclose(got[1], ref[1], rtol=1e-10)
    close_to_reference(got[0], g["T"][1])


@pytest.mark.parametrize("prune", [False, True])
def test_sub_range_of_a_packed_buffer(icp, prune, prec):
    """offsets that start in the middle of a larger packed buffer (off[0] > 0): the pruned path keeps its sorted
    copies in batch-local scratch, so a sub-range must give the full batch's answers and touch nothing else"""
    from lidar_slam_amd.batch import icp_batch
    pack_clouds = prec.pack
    rng = np.random.default_rng(321)
    sizes = [(900, 1100), (4096, 4096), (300, 2000), (1081, 1081), (2000, 64), (64, 3000), (1500, 1500), (700, 4096),
             (4096, 100), (1200, 1000)]
    ss, ts = zip(*[_random_problem(rng, 2, n, m, spread=rng.uniform(1.0, 15.0)) for n, m in sizes])
    fs, so = pack_clouds(list(ss))
    ft, to = pack_clouds(list(ts))
    I = np.tile(np.eye(4), (len(sizes), 1, 1))
    kw = dict(epsilon=1e-12, max_iters=8, return_correspondences=True)
    full = icp_batch(fs, so, ft, to, I, prune=False, **kw)
    for lo, hi in ((6, 10), (3, 4), (9, 10)):
        sub = icp_batch(fs, so[lo:hi + 1], ft, to[lo:hi + 1], I[lo:hi], prune=prune, **kw)
        np.testing.assert_array_equal(sub.iterations, full.iterations[lo:hi])
        np.testing.assert_array_equal(sub.correspondences[so[lo]:so[hi]], full.correspondences[so[lo]:so[hi]])
        np.testing.assert_allclose(sub.T, full.T[lo:hi], atol=1e-10)


def _oracle_check(r, srcs, tgts, so, init, recipe, **kw):
    from oracle import c_oracle
    for k, (s, t) in enumerate(zip(srcs, tgts)):
        To, eo, co, ito = c_oracle.run_icp(s, t, init[k], recipe=recipe, **kw)
        assert r.iterations[k] == ito, k
        np.testing.assert_array_equal(r.correspondences[so[k]:so[k + 1]], co, err_msg=f"problem {k}")
        np.testing.assert_allclose(r.T[k], To, atol=1e-9, err_msg=f"problem {k}")


@pytest.mark.parametrize("dim", [2, 3])
@pytest.mark.parametrize("ppl", ["1", "2", "4", "8", "auto"])
def test_filter_sweep_gives_the_exact_answers(icp, dim, ppl, monkeypatch, prec):
    """the big build's expanded-form filter (|t|² − 2a·t, then the exact recipe on the chunks within the rounding
    bound) against the C oracle's exact device recipe: random clouds at several scales, duplicates, clouds far from
    the origin (bound too loose: the wave must fall back), tiny and ragged sizes — and against the same build with the filter off"""
    from lidar_slam_amd.batch import icp_batch
    pack_clouds = prec.pack
    if ppl == "8" and (dim == 3 or prec.name == "f64"):
        pytest.skip("eight points per lane exist for the planar float32 build only")
    monkeypatch.setenv("LSLAM_ICP_VARIANT", "1")            # the big build whatever the size
    if ppl != "auto":                                        # auto: the launcher's own choice (8 for big planar clouds)
        monkeypatch.setenv("LSLAM_ICP_POINTS_PER_LANE", ppl)
    rng = np.random.default_rng(7000 + dim)
    sizes = [(3000, 4000), (8, 6), (70, 33), (4096, 4096), (1500, 31), (1000, 32), (2000, 2049), (640, 3000), (1300, 700)]
    if dim == 3:
        sizes = [(n, min(m, 4096)) for n, m in sizes]
    ss, ts = [], []
    for k, (n, m) in enumerate(sizes):
        s_, t_ = _random_problem(rng, dim, n, m, spread=[0.05, 1.0, 10.0, 300.0][k % 4])
        if m > 40:
            t_[m // 2] = t_[3]; t_[m - 1] = t_[3]           # exact duplicates in different chunks
        if k == 6:
            s_ += 5000.0; t_ += 5000.0                       # far from the origin: |t|² − 2a·t loses every digit that matters
        if k == 7:
            s_[:, 0] += 80.0; t_[:, 0] += 80.0
        if k == 8:
            # 5 % of the targets exist three times, in three different chunks: the sources next to them have three chunk
            # minima inside the rounding window — a few per wave (searched one by one by the wave), in some waves enough
            # of them to send the wave to the exact sweep
            t_[300:335] = t_[:35]; t_[650:685] = t_[:35]
        ss.append(s_); ts.append(t_)
    fs, so = pack_clouds(ss)
    ft, to = pack_clouds(ts)
    I = np.tile(np.eye(4), (len(sizes), 1, 1))
    kw = dict(epsilon=1e-12, max_iters=6)
    r = icp_batch(fs, so, ft, to, I, return_correspondences=True, prune=False, **kw)
    _oracle_check(r, ss, ts, so, I, prec.recipe, **kw)
    monkeypatch.setenv("LSLAM_ICP_FILTER", "0")
    e = icp_batch(fs, so, ft, to, I, return_correspondences=True, prune=False, **kw)
    np.testing.assert_array_equal(r.correspondences, e.correspondences)
    np.testing.assert_array_equal(r.iterations, e.iterations)
    np.testing.assert_allclose(r.T, e.T, rtol=0, atol=0 if ppl != "auto" else 1e-11)   # same pairs and (same P) the same float64 summation order: identical bits


def test_filter_sweep_on_lattices_and_collapsed_clouds(icp, monkeypatch, prec):
    """massive ties (every chunk minimum equal): the filter must hand over to the exact sweep, lowest index wins"""
    from lidar_slam_amd.batch import icp_batch
    pack_clouds = prec.pack
    monkeypatch.setenv("LSLAM_ICP_VARIANT", "1")
    rng = np.random.default_rng(11)
    gx, gy = np.meshgrid(np.arange(60) * 0.25, np.arange(60) * 0.25)
    lattice = np.column_stack((gx.ravel(), gy.ravel()))                      # 3600 points
    shuffled = lattice[rng.permutation(lattice.shape[0])]
    cases = [
        (lattice + 0.125, shuffled),
        (shuffled[:3000], np.vstack((lattice[:2000], lattice[:2000][::-1]))),
        (rng.normal(size=(2700, 2)), np.tile([[1.5, -2.0]], (300, 1))),
        (np.tile([[0.3, 0.3]], (2600, 1)), lattice),
        (lattice, lattice),
        (rng.normal(size=(3000, 2)) * 1e-3, rng.normal(size=(3000, 2)) * 1e-3),      # sub-millimetre cloud
        (rng.normal(size=(2600, 2)) * 1e-18, rng.normal(size=(2600, 2)) * 1e-18),    # squares in the subnormal range
    ]
    fs, so = pack_clouds([c[0] for c in cases])
    ft, to = pack_clouds([c[1] for c in cases])
    I = np.tile(np.eye(4), (len(cases), 1, 1))
    kw = dict(epsilon=1e-30, max_iters=2, return_correspondences=True, prune=False)
    a = icp_batch(fs, so, ft, to, I, **kw)
    monkeypatch.setenv("LSLAM_ICP_FILTER", "0")
    b = icp_batch(fs, so, ft, to, I, **kw)
    np.testing.assert_array_equal(a.correspondences, b.correspondences)
    np.testing.assert_array_equal(a.iterations, b.iterations)
    np.testing.assert_array_equal(a.T, b.T)
    # first iteration of case 0 against a direct evaluation: lowest index among the 4 equidistant corners
    monkeypatch.delenv("LSLAM_ICP_FILTER")
    s32, t32 = prec.cast(lattice + 0.125), prec.cast(shuffled)
    one = icp_batch(s32, [0, 3600], t32, [0, 3600], np.eye(4)[None], epsilon=0.0, max_iters=0, stopping_thresh=0.0, return_correspondences=True, prune=False)
    d = ((s32[:, None, :] - t32[None, :, :]) ** 2).sum(-1)
    np.testing.assert_array_equal(one.correspondences, np.argmin(d, axis=1))


@pytest.mark.parametrize("dim", [2, 3])
@pytest.mark.parametrize("case", ["duplicates", "lattice", "far", "tiny", "ragged_tail"])
def test_whole_chip_filter_search_adversarial(icp, case, dim, monkeypatch, prec):
    """the whole-chip search sweeps the expanded form first (nn_tiled_filter_kernel, planar and 3-D): exact duplicates in
    different chunks and splits, lattices (every wave must re-sweep with the exact recipe), a cloud far from the origin
    (bound too loose), sub-millimetre clouds, target counts that leave ragged chunks — all must equal the C oracle's
    exact recipe, and the same search with the filter off"""
    from lidar_slam_amd._lib import check, lib, ptr
    from oracle import c_oracle
    monkeypatch.setenv("LSLAM_ICP_FUSE_TAIL", "0")           # three launches: the non-fused search kernels at every size
    rng = np.random.default_rng({"duplicates": 1, "lattice": 2, "far": 3, "tiny": 4, "ragged_tail": 5}[case] + 10 * dim)
    n, m = (9000 if dim == 3 else 66000), 20011             # the filter search kernel: from two (3-D) / four (planar) source points per lane
    if case == "lattice":
        side = 27 if dim == 3 else 140
        g = np.stack(np.meshgrid(*([np.arange(side)] * dim)), -1).reshape(-1, dim) * 0.125
        t = g[rng.permutation(g.shape[0])]
        s = g[rng.integers(0, g.shape[0], n)] + 0.0625
    else:
        s, t = _random_problem(rng, dim, n, m if case != "ragged_tail" else 4099, spread={"tiny": 1e-3}.get(case, 2.0))
        if case == "duplicates":
            t[5000:5040] = t[:40]; t[19000:19040] = t[:40]
        if case == "far":
            s += 3000.0; t += 3000.0
    s32, t32 = prec.cast(s), prec.cast(t)
    To, eo, co, ito = c_oracle.run_icp(s, t, np.eye(4), epsilon=1e-12, max_iters=3, recipe=prec.recipe)
    entry = lib().lslam_icp_large_f64 if prec.name == "f64" else lib().lslam_icp_large
    got = {}
    for flt in ("1", "0"):
        monkeypatch.setenv("LSLAM_ICP_FILTER", flt)
        T = np.empty((4, 4)); err = np.empty(1); it = np.empty(1, np.int32); corr = np.empty(s32.shape[0], np.int32)
        check(entry(ptr(s32), s32.shape[0], ptr(t32), t32.shape[0], dim, ptr(np.eye(4)), 1e-12, 1e-4, 3, 4, ptr(T), ptr(err), ptr(it), ptr(corr), 0))
        assert it[0] == ito
        np.testing.assert_array_equal(corr, co)
        # 3000 m from the origin the float64 moments themselves lose digits to cancellation (order of summation differs)
        np.testing.assert_allclose(T, To, atol=1e-5 if case == "far" else 1e-9)
        got[flt] = (T.copy(), corr.copy())
    np.testing.assert_array_equal(got["1"][1], got["0"][1])
    np.testing.assert_allclose(got["1"][0], got["0"][0], atol=1e-5 if case == "far" else 1e-12)


def test_device_entry_points_on_alternating_streams(icp, prec):
    if prec.name == "f64":
        pytest.skip("workspace ordering does not depend on the precision mode")
    """the thread's workspaces (sorted copies, key rows, iteration state) are reused from call to call; calls on different
    streams are ordered by the library (an event per call), so interleaving streams changes nothing"""
    import torch
    from lidar_slam_amd.batch import icp_batch_dev, icp_large_dev
    from lidar_slam_amd.synth import batch_c4
    dev = torch.device("cuda", 0)
    w = batch_c4(64, 1024, seed=3)
    src = torch.from_numpy(w.src).to(dev); tgt = torch.from_numpy(w.tgt).to(dev)
    off = torch.from_numpy(w.src_off).to(dev); init = torch.from_numpy(w.init_T).to(dev)
    S = 64
    streams = [torch.cuda.Stream(dev), torch.cuda.Stream(dev), torch.cuda.current_stream(dev)]
    rng = np.random.default_rng(0)
    big_s = torch.from_numpy(rng.normal(size=(9000, 3)).astype(np.float32)).to(dev)
    big_t = torch.from_numpy(rng.normal(size=(30000, 3)).astype(np.float32)).to(dev)
    torch.cuda.synchronize()

    def batch_on(stream, prune):
        oT = torch.empty((S, 4, 4), dtype=torch.float64, device=dev); oe = torch.empty(S, dtype=torch.float64, device=dev)
        oi = torch.empty(S, dtype=torch.int32, device=dev)
        icp_batch_dev(src.data_ptr(), off.data_ptr(), tgt.data_ptr(), off.data_ptr(), 2, init.data_ptr(), S, 1024, 1024,
                      oT.data_ptr(), oe.data_ptr(), oi.data_ptr(), None, epsilon=0.0, max_iters=5, stopping_thresh=0.0,
                      stream=stream.cuda_stream, prune=prune)
        return oT, oi

    ref = {}
    for prune in (False, True):
        oT, oi = batch_on(streams[2], prune)
        torch.cuda.synchronize()
        ref[prune] = oT.cpu().numpy()
    ref_large = icp_large_dev(big_s.data_ptr(), 9000, big_t.data_ptr(), 30000, 3, np.eye(4), epsilon=0.0, max_iters=3, stopping_thresh=0.0,
                              stream=streams[2].cuda_stream, prune=False)[0]
    outs = []
    for k in range(12):
        st = streams[k % 3]
        prune = bool(k & 1)
        outs.append((prune, batch_on(st, prune)[0]))
        if k % 4 == 0:
            T = icp_large_dev(big_s.data_ptr(), 9000, big_t.data_ptr(), 30000, 3, np.eye(4), epsilon=0.0, max_iters=3, stopping_thresh=0.0,
                              stream=streams[(k + 1) % 3].cuda_stream, prune=False)[0]
            np.testing.assert_array_equal(T, ref_large)
    torch.cuda.synchronize()
    for prune, oT in outs:
        np.testing.assert_array_equal(oT.cpu().numpy(), ref[prune])


def test_headline_batch_shape_eight_points_per_lane(icp, monkeypatch, prec):
    if prec.name == "f64":
        pytest.skip("the eight-points-per-lane build is the float32 headline kernel")
    """2048 problems of 4096 x 4096 planar points all pairs: the launcher's own choice there is the filter build with
    EIGHT source points per lane in 8-wave workgroups (bench.py's headline kernel).  Sampled problems equal the C oracle's
    exact recipe; the whole batch equals the four-points-per-lane build."""
    from lidar_slam_amd.batch import icp_batch
    from lidar_slam_amd.synth import batch_c4
    from oracle import c_oracle
    S = 2048
    w = batch_c4(S, 4096, seed=9)
    kw = dict(epsilon=1e-3, return_correspondences=True, prune=False)
    r = icp_batch(w.src, w.src_off, w.tgt, w.tgt_off, w.init_T, **kw)
    for k in (0, 777, S - 1):
        sl = slice(k * 4096, (k + 1) * 4096)
        To, eo, co, ito = c_oracle.run_icp(w.src[sl].astype(np.float64), w.tgt[sl].astype(np.float64), w.init_T[k], epsilon=1e-3, recipe="device")
        assert r.iterations[k] == ito
        np.testing.assert_array_equal(r.correspondences[sl], co)
        np.testing.assert_allclose(r.T[k], To, atol=1e-9)
    monkeypatch.setenv("LSLAM_ICP_POINTS_PER_LANE", "4")
    r4 = icp_batch(w.src, w.src_off, w.tgt, w.tgt_off, w.init_T, **kw)
    np.testing.assert_array_equal(r.iterations, r4.iterations)
    np.testing.assert_array_equal(r.correspondences, r4.correspondences)
    np.testing.assert_allclose(r.T, r4.T, atol=1e-11)
    err_t = np.abs(r.T[:, :2, 3] - w.true_T[:, :2, 3]).max()
    assert err_t < 2e-3, err_t                                        # 1 cm noise on 4096 points


@pytest.mark.parametrize("strategy", ["filter", "direct", "pruned"])
def test_full_size_dense_3d_identity_search(icp, strategy, monkeypatch, prec):
    if prec.name == "f64":
        pytest.skip("device-pointer float32 search; the float64-exact whole-chip path has its own full-size test below")
    """BASELINE's 100k x 100k 3-D size through a size-independent property: every point of a cloud is its own nearest
    neighbour in that cloud, at distance exactly 0 — on the filter search, the direct-difference search and the pruned
    search; and one ICP iteration from the identity leaves the identity."""
    import torch
    from lidar_slam_amd.batch import icp_large_dev, nn_dev
    from lidar_slam_amd.synth import cloud_c5
    _, tgt, _, _ = cloud_c5()
    assert tgt.shape == (100_000, 3)
    assert np.unique(tgt, axis=0).shape[0] == tgt.shape[0]             # no duplicate points: the identity is the unique answer
    d_t = torch.from_numpy(tgt).cuda()
    st = torch.cuda.current_stream().cuda_stream
    if strategy == "direct":
        monkeypatch.setenv("LSLAM_ICP_FILTER", "0")
    if strategy != "pruned":
        idx = torch.empty(tgt.shape[0], dtype=torch.int32, device="cuda")
        d2 = torch.empty(tgt.shape[0], dtype=torch.float32, device="cuda")
        nn_dev(d_t.data_ptr(), tgt.shape[0], d_t.data_ptr(), tgt.shape[0], 3, None, idx.data_ptr(), d2.data_ptr(), stream=st)
        torch.cuda.synchronize()
        np.testing.assert_array_equal(idx.cpu().numpy(), np.arange(tgt.shape[0], dtype=np.int32))
        assert float(d2.max().item()) == 0.0
    corr = torch.empty(tgt.shape[0], dtype=torch.int32, device="cuda")
    T, err, it = icp_large_dev(d_t.data_ptr(), tgt.shape[0], d_t.data_ptr(), tgt.shape[0], 3, np.eye(4), epsilon=0.0, max_iters=0,
                               stopping_thresh=0.0, d_out_corr=corr.data_ptr(), stream=st, prune=(strategy == "pruned"))
    torch.cuda.synchronize()
    assert it == 1 and err == 0.0
    np.testing.assert_array_equal(corr.cpu().numpy(), np.arange(tgt.shape[0], dtype=np.int32))
    np.testing.assert_allclose(T, np.eye(4), atol=1e-12)


@pytest.mark.parametrize("plist", ["1", "0"])
def test_exact_pruning_3d_gives_the_brute_force_answers(icp, prec, plist, monkeypatch):
    """(both ways of evaluating the surviving chunks: per-lane lists, the default, and the whole wave)
    the 3-D pruned batch kernel (Morton-sorted float4 copies, 16³ grid, chunk boxes in 3-D): identical correspondences and
    iteration counts as the all-pairs sweep and as the C oracle's exact recipe — ragged sizes up to the kernel's limits
    (4096 targets, 8191 sources), duplicates far apart in index, a lattice with massive ties, a collapsed target,
    normalised error, a sub-range of a packed buffer"""
    monkeypatch.setenv("LSLAM_ICP_PRUNE_LIST", plist)
    from lidar_slam_amd.batch import icp_batch
    pack_clouds = prec.pack
    from oracle import c_oracle
    rng = np.random.default_rng(333)
    sizes = [(8, 6), (100, 3000), (4096, 4096), (3000, 700), (8191, 64), (513, 4096), (6000, 4000), (1500, 1500)]
    ss, ts = [], []
    for n, m in sizes:
        s_, t_ = _random_problem(rng, 3, n, m, spread=rng.uniform(0.05, 20.0))
        if m > 40:
            t_[m // 2] = t_[3]; t_[m - 1] = t_[3]          # exact duplicates far apart in index
        ss.append(s_); ts.append(t_)
    g = np.stack(np.meshgrid(np.arange(14), np.arange(14), np.arange(14)), -1).reshape(-1, 3) * 0.25
    ss.append(g[rng.integers(0, g.shape[0], 2000)] + 0.125); ts.append(g[rng.permutation(g.shape[0])])     # 8-way ties everywhere
    # (a target collapsed to ONE point is left out in 3-D: its cross-covariance is rounding noise and the rotation, hence the
    #  iteration count, follows the summation order — parity undefined in the reference too)
    fs, so = pack_clouds(ss)
    ft, to = pack_clouds(ts)
    I = np.tile(np.eye(4), (len(ss), 1, 1))
    for norm in (False, True):
        kw = dict(epsilon=1e-12, max_iters=8, normalize_error=norm, return_correspondences=True)
        a = icp_batch(fs, so, ft, to, I, prune=False, **kw)
        b = icp_batch(fs, so, ft, to, I, prune=True, **kw)
        np.testing.assert_array_equal(b.iterations, a.iterations)
        np.testing.assert_array_equal(b.correspondences, a.correspondences)
        defined = list(range(1, len(ss)))                     # problem 0: 8 sources on 6 targets may match two of them only — rotation not unique
        assert np.isfinite(a.T[defined]).all() and np.isfinite(b.T[defined]).all()
        np.testing.assert_allclose(b.T[defined], a.T[defined], atol=1e-9)
    c = icp_batch(fs, so, ft, to, I, prune=True, epsilon=1e-12, max_iters=8, return_correspondences=True)
    for k in (2, 5, 6):
        To, eo, co, ito = c_oracle.run_icp(ss[k], ts[k], np.eye(4), epsilon=1e-12, max_iters=8, recipe=prec.recipe)
        assert c.iterations[k] == ito
        np.testing.assert_array_equal(c.correspondences[so[k]:so[k + 1]], co)
        np.testing.assert_allclose(c.T[k], To, atol=1e-9)
    sub = icp_batch(fs, so[5:8], ft, to[5:8], I[5:7], prune=True, epsilon=1e-12, max_iters=8, return_correspondences=True)
    np.testing.assert_array_equal(sub.correspondences[so[5]:so[7]], c.correspondences[so[5]:so[7]])
    np.testing.assert_allclose(sub.T, c.T[5:7], atol=1e-10)


_C3 = {}


def _c3_sequence():
    """the dataset-20-like sequence of BASELINE configs[2], generated once per session"""
    if "seq" not in _C3:
        from lidar_slam_amd.synth import sequence_c3
        _C3["seq"] = sequence_c3(4956, seed=20)
    return _C3["seq"]


def test_full_size_c3_sequence_vs_oracle(icp, prec):
    """BASELINE configs[2] at FULL size: 4956 scans (4955 ragged pairs, 650-1081 points) through poses_from_scan_matching —
    one launch.  400 pairs spread over the sequence are re-run by the C oracle: identical iteration counts, relative poses
    within 1e-9 of the oracle's; the chain of all 4955 relative poses equals the oracle's sequential float64 chain."""
    from lidar_slam_amd.modules.localization import (chain_poses, poses_from_scan_matching, relative_pose_guesses,
                                                     scan_matching_relative_poses)
    from oracle import c_oracle, se2_oracle
    seq = _c3_sequence()
    z = [prec.cast(a) for a in seq.z_ts]
    rel, info = scan_matching_relative_poses(seq.x_ts, z, return_info=True)
    assert rel.shape == (4955, 3, 3) and info.iterations.shape == (4955,) and info.iterations.min() >= 1
    init = relative_pose_guesses(seq.x_ts)
    sample = np.linspace(0, 4954, 400).astype(int)
    for k in sample:
        To, eo, co, ito = c_oracle.run_icp(seq.z_ts[k + 1], seq.z_ts[k], init[k], epsilon=0.001, recipe=prec.recipe)
        assert info.iterations[k] == ito, k
        assert np.abs(rel[k][:2, :2] - To[:2, :2]).max() < 1e-9 and np.abs(rel[k][:2, 2] - To[:2, 3]).max() < 1e-9, k
        assert abs(info.error[k] - eo) < 1e-9 * max(1.0, eo)
    poses = poses_from_scan_matching(seq.x_ts, z)
    np.testing.assert_array_equal(poses, chain_poses(rel))
    np.testing.assert_allclose(poses, se2_oracle.chain_relative_poses(rel), atol=1e-9)


def test_full_size_c4_batch_vs_oracle(icp, prec):
    """BASELINE configs[3] at FULL size: 4096 pairs x 4096 x 4096 points in one launch (the library's default search
    strategy, reference stop rules).  Three sampled problems equal the C oracle (iterations, every correspondence, T);
    the batch recovers the generating motions; the all-pairs strategy gives the same iteration counts for every pair."""
    from lidar_slam_amd.batch import icp_batch
    from lidar_slam_amd.synth import batch_c4
    from oracle import c_oracle
    S = 4096
    w = batch_c4(S, 4096, seed=11)
    src, tgt = prec.cast(w.src), prec.cast(w.tgt)
    r = icp_batch(src, w.src_off, tgt, w.tgt_off, w.init_T, epsilon=1e-3, return_correspondences=True)
    assert r.iterations.shape == (S,) and r.iterations.min() >= 1
    for k in (0, 1777, S - 1):
        sl = slice(k * 4096, (k + 1) * 4096)
        To, eo, co, ito = c_oracle.run_icp(w.src[sl].astype(np.float64), w.tgt[sl].astype(np.float64), w.init_T[k], epsilon=1e-3, recipe=prec.recipe)
        assert r.iterations[k] == ito
        np.testing.assert_array_equal(r.correspondences[sl], co)
        np.testing.assert_allclose(r.T[k], To, atol=1e-9)
    assert np.abs(r.T[:, :2, 3] - w.true_T[:, :2, 3]).max() < 2e-3            # 1 cm noise on 4096 points
    a = icp_batch(src, w.src_off, tgt, w.tgt_off, w.init_T, epsilon=1e-3, prune=False)
    np.testing.assert_array_equal(a.iterations, r.iterations)
    np.testing.assert_allclose(a.T, r.T, atol=1e-10)


@pytest.mark.parametrize("dim", [2, 3])
def test_filter_bound_with_sources_far_from_a_target_at_the_origin(icp, dim, monkeypatch, prec):
    """advisor, round 2: the d² coefficient of the filter's window (9u planar, 11u in 3-D) matters when d² ≫ R(R + |a|) is
    NOT dominated by the E term — a target cloud near the origin (R ≈ 1) and sources 10–40 R away.  Batched filter build and
    whole-chip filter search against the C oracle and against the same builds with the filter off.  ONE iteration from the
    identity, so that the search really runs from far away; with every source matched to the few targets facing it the
    cross-covariance is nearly rank-deficient and the rotation noise-dominated (parity undefined): correspondences
    and error are compared, not T."""
    from lidar_slam_amd.batch import icp_batch
    from lidar_slam_amd._lib import check, lib, ptr
    from oracle import c_oracle
    rng = np.random.default_rng(900 + dim)
    monkeypatch.setenv("LSLAM_ICP_VARIANT", "1")
    cases = []
    for n, m, shift in ((3000, 4000, 12.0), (2500, 2048, 40.0), (700, 333, 25.0)):
        t = rng.normal(size=(m, dim)) * 0.4                       # R ≈ 1.5
        s = rng.normal(size=(n, dim)) * 0.4
        s[:, 0] += shift                                          # ≥ 10 R away
        t[m // 3] = t[7]                                          # an exact duplicate
        cases.append((s, t))
    fs, so = prec.pack([c[0] for c in cases]); ft, to = prec.pack([c[1] for c in cases])
    I = np.tile(np.eye(4), (len(cases), 1, 1))
    kw = dict(epsilon=1e-12, max_iters=0, return_correspondences=True, prune=False)
    a = icp_batch(fs, so, ft, to, I, **kw)
    for k, (s_, t_) in enumerate(cases):
        To, eo, co, ito = c_oracle.run_icp(s_, t_, np.eye(4), epsilon=1e-12, max_iters=0, recipe=prec.recipe)
        assert a.iterations[k] == ito == 1
        np.testing.assert_array_equal(a.correspondences[so[k]:so[k + 1]], co)
        np.testing.assert_allclose(a.error[k], eo, rtol=1e-9)
    monkeypatch.setenv("LSLAM_ICP_FILTER", "0")
    b = icp_batch(fs, so, ft, to, I, **kw)
    np.testing.assert_array_equal(a.correspondences, b.correspondences)
    np.testing.assert_array_equal(a.error, b.error)
    monkeypatch.delenv("LSLAM_ICP_FILTER")
    # whole-chip filter search (two / four source points per lane), three launches per iteration
    monkeypatch.setenv("LSLAM_ICP_FUSE_TAIL", "0")
    n, m = (9000 if dim == 3 else 66000), 6007
    t = rng.normal(size=(m, dim)) * 0.4
    s = rng.normal(size=(n, dim)) * 0.4
    s[:, 1] -= 30.0
    entry = lib().lslam_icp_large_f64 if prec.name == "f64" else lib().lslam_icp_large
    To, eo, co, ito = c_oracle.run_icp(s, t, np.eye(4), epsilon=1e-12, max_iters=0, recipe=prec.recipe)
    sc, tc = prec.cast(s), prec.cast(t)
    T = np.empty((4, 4)); err = np.empty(1); it = np.empty(1, np.int32); corr = np.empty(n, np.int32)
    check(entry(ptr(sc), n, ptr(tc), m, dim, ptr(np.eye(4)), 1e-12, 1e-4, 0, 4, ptr(T), ptr(err), ptr(it), ptr(corr), 0))
    assert it[0] == ito == 1
    np.testing.assert_array_equal(corr, co)
    np.testing.assert_allclose(err[0], eo, rtol=1e-9)


@pytest.mark.parametrize("dim", [2, 3])
def test_float64_exact_mode_on_clouds_made_of_near_ties(icp, dim, monkeypatch, prec):
    """the worst case for float64-exact mode: targets on a lattice moved by 1e-9 … 1e-6 (far below float32 resolution at these
    coordinates), sources at the cell centres moved likewise — EVERY source has 2^dim candidates the float32 search cannot tell
    apart, spread over several chunks, and the float64 winner is decided by the nanometre perturbations.  All batched kernels
    (all pairs: small / big / filter builds; pruned), the co-resident kernel and both whole-chip paths against the C oracle's
    float64 recipe.  (float32 clouds: the same data rounded — massive exact ties, lowest index — against the device recipe.)"""
    from lidar_slam_amd import _lib
    from lidar_slam_amd._lib import check, lib, ptr
    from lidar_slam_amd.batch import icp_batch
    from oracle import c_oracle
    rng = np.random.default_rng(4242 + dim)
    side = 40 if dim == 2 else 12
    g = np.stack(np.meshgrid(*([np.arange(side)] * dim)), -1).reshape(-1, dim) * 0.5 + 7.0          # coordinates 7 … 27: float32 ulp ≈ 1e-6
    cases = []
    for scale in (1e-9, 1e-7, 1e-6):
        t = g[rng.permutation(g.shape[0])] + rng.uniform(-1, 1, g.shape) * scale
        s = g[rng.integers(0, g.shape[0], 900)] + 0.25 + rng.uniform(-1, 1, (900, dim)) * scale
        cases.append((s, t))
    fs, so = prec.pack([c[0] for c in cases]); ft, to = prec.pack([c[1] for c in cases])
    I = np.tile(np.eye(4), (len(cases), 1, 1))
    kw = dict(epsilon=0.0, stopping_thresh=0.0, max_iters=1, return_correspondences=True)
    oracles = [c_oracle.run_icp(s, t, np.eye(4), epsilon=0.0, stopping_thresh=0.0, max_iters=1, recipe=prec.recipe) for s, t in cases]
    for prune, variant in ((False, None), (False, "1"), (True, None)):
        if variant:
            monkeypatch.setenv("LSLAM_ICP_VARIANT", variant)
        r = icp_batch(fs, so, ft, to, I, prune=prune, **kw)
        monkeypatch.delenv("LSLAM_ICP_VARIANT", raising=False)
        for k, (To, eo, co, ito) in enumerate(oracles):
            assert r.iterations[k] == ito == 2
            np.testing.assert_array_equal(r.correspondences[so[k]:so[k + 1]], co, err_msg=f"prune={prune} variant={variant} case {k}")
            np.testing.assert_allclose(r.T[k], To, atol=1e-9)
    entry = lib().lslam_icp_large_f64 if prec.name == "f64" else lib().lslam_icp_large
    for k, (s, t) in enumerate(cases):
        sc, tc = prec.cast(s), prec.cast(t)
        for flags, coop in ((_lib.ICP_ALL_PAIRS, "1"), (_lib.ICP_ALL_PAIRS, "0"), (_lib.ICP_PRUNE, "1")):
            monkeypatch.setenv("LSLAM_ICP_COOP", coop)
            T = np.empty((4, 4)); err = np.empty(1); it = np.empty(1, np.int32); corr = np.empty(s.shape[0], np.int32)
            check(entry(ptr(sc), s.shape[0], ptr(tc), t.shape[0], dim, ptr(np.eye(4)), 0.0, 0.0, 1, flags, ptr(T), ptr(err), ptr(it), ptr(corr), 0))
            assert it[0] == 2
            np.testing.assert_array_equal(corr, oracles[k][2], err_msg=f"whole-chip flags={flags} coop={coop} case {k}")
            np.testing.assert_allclose(T, oracles[k][0], atol=1e-9)


# ---- the co-resident one-launch kernel must degrade, never fail (lslam.h: lslam_icp_coop_fallbacks) ------------------------
def _c2_like_pair(seed=5, n=1081):
    rng = np.random.default_rng(seed)
    ang = np.linspace(-2.35, 2.35, n)
    rad = 5.0 + np.sin(3 * ang) + 0.6 * np.cos(7 * ang) + rng.normal(0, 0.004, n)
    tgt = np.column_stack((rad * np.cos(ang), rad * np.sin(ang)))
    th, t = 0.03, np.array([0.05, 0.01])
    R = np.array([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]])
    return (tgt - t) @ R + rng.normal(0, 0.003, tgt.shape), tgt


def test_co_resident_launch_falls_back_instead_of_failing(icp, prec, monkeypatch):
    """a hand-off bound of zero makes the first wait of the co-resident kernel give up: the SAME call must then finish on the
    launch-per-iteration path with the same result; a CU count too small for the grid must skip the co-resident path"""
    from lidar_slam_amd import _lib
    src, tgt = _c2_like_pair()
    src, tgt = prec.cast(src), prec.cast(tgt)
    monkeypatch.setenv("LSLAM_ICP_PRUNE", "0")
    ref = icp.run_icp(src, tgt, epsilon=0.0, stopping_thresh=0.0, max_iters=11, return_error=True, return_correspondences=True)
    before = _lib.lib().lslam_icp_coop_fallbacks()
    monkeypatch.setenv("LSLAM_ICP_COOP_TIMEOUT_US", "0")
    got = icp.run_icp(src, tgt, epsilon=0.0, stopping_thresh=0.0, max_iters=11, return_error=True, return_correspondences=True)
    monkeypatch.delenv("LSLAM_ICP_COOP_TIMEOUT_US")
    assert _lib.lib().lslam_icp_coop_fallbacks() == before + 1, "the zero bound did not force the detour"
    assert np.abs(got[0] - ref[0]).max() < 1e-9 and abs(got[1] - ref[1]) < 1e-9 * max(1.0, ref[1])
    assert np.array_equal(got[2], ref[2])
    # a device with fewer CUs than twice the grid (17 workgroups here): the co-resident path is not taken at all
    monkeypatch.setenv("LSLAM_ICP_COOP_MAX_CUS", "32")
    monkeypatch.setenv("LSLAM_ICP_COOP_TIMEOUT_US", "0")          # would force a detour if the path were taken
    got2 = icp.run_icp(src, tgt, epsilon=0.0, stopping_thresh=0.0, max_iters=11, return_error=True, return_correspondences=True)
    assert _lib.lib().lslam_icp_coop_fallbacks() == before + 1
    assert np.abs(got2[0] - ref[0]).max() < 1e-9 and np.array_equal(got2[2], ref[2])


def test_single_pair_while_another_stream_holds_the_chip(icp, prec):
    """a long all-pairs batch occupies every CU from one stream while run_icp's device entry point solves a LiDAR-sized pair on
    another: whichever path the library ends on, the answer is the uncontended one (and the oracle's)"""
    import torch
    from lidar_slam_amd import batch as B
    from oracle import icp_oracle
    src, tgt = _c2_like_pair(seed=9)
    dev = torch.device("cuda", 0)
    d_src32 = torch.from_numpy(src.astype(np.float32)).to(dev); d_tgt32 = torch.from_numpy(tgt.astype(np.float32)).to(dev)
    d_src64 = torch.from_numpy(src).to(dev) if prec.name == "f64" else None
    d_tgt64 = torch.from_numpy(tgt).to(dev) if prec.name == "f64" else None
    if prec.name == "f64":
        d_src32 = d_src64.float(); d_tgt32 = d_tgt64.float()
    n = src.shape[0]
    kw = dict(epsilon=1e-3, prune=False, d_src64=d_src64.data_ptr() if d_src64 is not None else None,
              d_tgt64=d_tgt64.data_ptr() if d_tgt64 is not None else None)
    quiet = B.icp_large_dev(d_src32.data_ptr(), n, d_tgt32.data_ptr(), n, 2, np.eye(4), **kw)
    # the load: 2048 problems of 2048 x 2048 points, 40 forced all-pairs iterations (tens of milliseconds of a full chip)
    S, P = 2048, 2048
    g = torch.Generator(device=dev); g.manual_seed(3)
    big = (torch.rand((S * P, 2), device=dev, generator=g) * 20 - 10).contiguous()
    off = (torch.arange(S + 1, device=dev, dtype=torch.int64) * P).contiguous()
    init = torch.eye(4, dtype=torch.float64, device=dev).repeat(S, 1, 1).contiguous()
    oT = torch.empty((S, 4, 4), dtype=torch.float64, device=dev); oe = torch.empty(S, dtype=torch.float64, device=dev)
    oi = torch.empty(S, dtype=torch.int32, device=dev)
    load_stream, pair_stream = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    B.icp_batch_dev(big.data_ptr(), off.data_ptr(), big.data_ptr(), off.data_ptr(), 2, init.data_ptr(), S, P, P, oT.data_ptr(),
                    oe.data_ptr(), oi.data_ptr(), None, epsilon=0.0, max_iters=39, stopping_thresh=0.0, stream=load_stream.cuda_stream, prune=False)
    busy = B.icp_large_dev(d_src32.data_ptr(), n, d_tgt32.data_ptr(), n, 2, np.eye(4), stream=pair_stream.cuda_stream, **kw)
    torch.cuda.synchronize()
    assert int(oi.min().item()) == 40                                   # the load really ran
    assert busy[2] == quiet[2] and np.abs(busy[0] - quiet[0]).max() < 1e-12 and abs(busy[1] - quiet[1]) < 1e-12 * max(1.0, quiet[1])
    ref = icp_oracle.run_icp_full(src if prec.name == "f64" else src.astype(np.float32).astype(np.float64),
                                  tgt if prec.name == "f64" else tgt.astype(np.float32).astype(np.float64),
                                  epsilon=1e-3, recipe=prec.recipe, nn="brute")
    assert busy[2] == ref.iterations and np.abs(busy[0] - ref.T).max() < 1e-9


def test_release_thread_resources_and_continue(icp, prec):
    """lslam_release_thread_resources hands every per-thread workspace back; the next calls allocate again and agree"""
    import lidar_slam_amd
    from lidar_slam_amd.batch import icp_batch
    src, tgt = _c2_like_pair(seed=2, n=700)
    clouds = [prec.cast(src), prec.cast(tgt)]
    flat, off = prec.pack(clouds)
    first = icp_batch(flat, np.array([off[0], off[1]]), flat, np.array([off[1], off[2]]), np.eye(4)[None], epsilon=1e-3, prune=True)
    one = icp.run_icp(clouds[0], clouds[1], epsilon=1e-3)
    lidar_slam_amd.release_thread_resources()
    lidar_slam_amd.release_thread_resources()                               # idempotent
    again = icp_batch(flat, np.array([off[0], off[1]]), flat, np.array([off[1], off[2]]), np.eye(4)[None], epsilon=1e-3, prune=True)
    one_again = icp.run_icp(clouds[0], clouds[1], epsilon=1e-3)
    assert np.array_equal(first.T, again.T) and np.array_equal(first.iterations, again.iterations)
    assert np.array_equal(one, one_again)


# ---- the task-queue kernel (csrc/lslam_icp_queue.hip): slots, tasks and matches kept on a proof ------------------------------------
@pytest.mark.parametrize("knobs", [{}, {"LSLAM_ICP_QUEUE_CERTIFY": "1"}, {"LSLAM_ICP_QUEUE_CERTIFY": "1", "LSLAM_ICP_QUEUE_Q": "1", "LSLAM_ICP_QUEUE_WAVES": "3", "LSLAM_ICP_QUEUE_G": "5"},
                                   {"LSLAM_ICP_QUEUE_CERTIFY": "0", "LSLAM_ICP_QUEUE_Q": "2", "LSLAM_ICP_QUEUE_WAVES": "16"}, {"LSLAM_ICP_QUEUE_CERTIFY": "1", "LSLAM_ICP_QUEUE_ECAP": "4.0"},
                                   {"LSLAM_ICP_QUEUE_CERTIFY": "1", "LSLAM_ICP_QUEUE_LEAN_BELOW": "99"}, {"LSLAM_ICP_QUEUE_CERTIFY": "1", "LSLAM_ICP_QUEUE_FUSE": "0"},
                                   {"LSLAM_ICP_QUEUE_INCR": "0"}, {"LSLAM_ICP_QUEUE_CERTIFY": "1", "LSLAM_ICP_QUEUE_INCR": "0", "LSLAM_ICP_QUEUE_G": "3"},
                                   {"LSLAM_ICP_QUEUE_CERTIFY": "1", "LSLAM_ICP_QUEUE_CONTESTED": "0", "LSLAM_ICP_QUEUE_E0": "0", "LSLAM_ICP_QUEUE_EFLOOR": "0"}])
def test_task_queue_kernel_gives_the_all_pairs_answers(icp, prec, knobs, monkeypatch):
    """ragged planar batch from one point to 4096 per cloud — more problems than a CU has slots, tiny and large ones mixed, duplicates, a
    lattice, motions far from the start, many forced iterations (so that most sources hold a certificate most of the time), the normalised
    error's bounding box — through the pruned path in several geometries (slots, waves, slabs per task, certificates forced on / off, a reach
    extension far beyond its default cap, sources read from memory instead of LDS, running sums on / off — by default the problems from
    2048 points per cloud run on double-double sums of changes and the smaller ones in a launch of their own): identical correspondences
    and iteration counts as the all-pairs sweep, T and error to float64 summation order"""
    from lidar_slam_amd.batch import icp_batch
    for k, v in knobs.items():
        monkeypatch.setenv(k, v)
    rng = np.random.default_rng(77)
    sizes = [(1, 1), (3, 70), (64, 64), (65, 17), (700, 1081), (1081, 650), (2048, 2048), (2100, 3000), (4096, 4096), (129, 4000), (3000, 40)] + [(900 + 7 * k, 1000 - 5 * k) for k in range(12)]
    srcs, tgts, inits = [], [], []
    for n, m in sizes:
        s, t = _random_problem(rng, 2, n, m, spread=rng.uniform(0.5, 15.0))
        if m > 40:
            t[m // 2] = t[3]                                  # an exact duplicate: lowest index wins
        srcs.append(s); tgts.append(t)
        th = rng.uniform(-0.3, 0.3)
        T0 = np.eye(4); T0[:2, :2] = [[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]]; T0[:2, 3] = rng.normal(0, 0.3, 2)
        inits.append(T0)
    gx, gy = np.meshgrid(np.arange(40.0), np.arange(30.0))                       # a lattice: massive exact ties, again and again
    lattice = np.column_stack((gx.ravel(), gy.ravel()))
    srcs.append(lattice[::2] + [0.5, 0.5]); tgts.append(lattice); inits.append(np.eye(4))
    fs, so = prec.pack(srcs)
    ft, to = prec.pack(tgts)
    I = np.stack(inits)
    for kw in (dict(epsilon=0.0, stopping_thresh=0.0, max_iters=29), dict(epsilon=1e-3, normalize_error=True), dict(epsilon=1e-6, stopping_thresh=1e-9, max_iters=60)):
        a = icp_batch(fs, so, ft, to, I, return_correspondences=True, prune=False, **kw)
        b = icp_batch(fs, so, ft, to, I, return_correspondences=True, prune=True, **kw)
        np.testing.assert_array_equal(b.iterations, a.iterations)
        np.testing.assert_array_equal(b.correspondences, a.correspondences)
        ok = np.isfinite(a.T).all(axis=(1, 2))
        np.testing.assert_allclose(b.T[ok], a.T[ok], atol=1e-9)
        np.testing.assert_allclose(b.error[ok], a.error[ok], rtol=1e-9, atol=1e-12)


@pytest.mark.parametrize("kw", [dict(epsilon=0.0, stopping_thresh=0.0, max_iters=0), dict(epsilon=1e9), dict(epsilon=0.0, stopping_thresh=0.0, max_iters=3),
                                dict(epsilon=1e-6, stopping_thresh=1e-12, max_iters=40)])
def test_running_sums_at_the_edges(icp, prec, kw):
    """the running-sums build (every problem here has a cloud of 2048 points or more) where its phases degenerate: a single iteration (the
    first round of sums and nothing else), a stop on the first error, one source, one target, a handful of targets (every source matched to
    one of five points), all targets the same point, more sources than targets and the reverse, 17 problems for 4 slots — against the all-pairs
    sweep: iteration counts, correspondences, T, error"""
    from lidar_slam_amd.batch import icp_batch
    rng = np.random.default_rng(2048)
    sizes = [(4096, 4096), (2048, 5), (5, 2048), (1, 4096), (4096, 1), (3000, 2100), (2049, 64), (64, 2049)] + [(2048 + 100 * k, 2300 - 20 * k) for k in range(8)]
    srcs, tgts, inits = [], [], []
    for n, m in sizes:
        s, t = _random_problem(rng, 2, n, m, spread=6.0)
        srcs.append(s); tgts.append(t)
        th = rng.uniform(-0.05, 0.05)
        T0 = np.eye(4); T0[:2, :2] = [[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]]; T0[:2, 3] = rng.normal(0, 0.05, 2)
        inits.append(T0)
    s, t = _random_problem(rng, 2, 2500, 2500, spread=6.0)
    t[:] = t[0]                                               # every target the same point: one massive exact tie, the lowest index wins
    srcs.append(s); tgts.append(t); inits.append(np.eye(4))
    fs, so = prec.pack(srcs); ft, to = prec.pack(tgts)
    I = np.stack(inits)
    a = icp_batch(fs, so, ft, to, I, return_correspondences=True, prune=False, **kw)
    b = icp_batch(fs, so, ft, to, I, return_correspondences=True, prune=True, **kw)
    np.testing.assert_array_equal(b.correspondences, a.correspondences)
    # every source matched to ONE point: the centred targets are zero to rounding, the rotation is whatever the noise says — in the reference
    # too (parity undefined); those two problems are compared on their correspondences only
    ok = np.isfinite(a.T).all(axis=(1, 2))
    ok[[4, len(sizes)]] = False
    np.testing.assert_array_equal(b.iterations[ok], a.iterations[ok])
    assert np.isfinite(b.T[ok]).all()
    np.testing.assert_allclose(b.T[ok], a.T[ok], atol=1e-9)
    np.testing.assert_allclose(b.error[ok], a.error[ok], rtol=1e-9, atol=1e-12)


def test_task_queue_kernel_says_so_when_waves_give_up(icp, monkeypatch):
    """the can't-happen guard of the task-queue kernel: waves that see no sign of life in their workgroup for 4 s leave, so that the grid
    always drains.  With the bound set to (almost) nothing every idle poll gives up: the launch ends, results are missing, and the host
    entry point must return an error instead of them; the next ordinary call works again."""
    from lidar_slam_amd._lib import LslamError
    from lidar_slam_amd.batch import icp_batch
    rng = np.random.default_rng(3)
    srcs, tgts = [], []
    for _ in range(600):
        s, t = _random_problem(rng, 2, 1081, 1081, spread=8.0)
        srcs.append(s.astype(np.float32)); tgts.append(t.astype(np.float32))
    from lidar_slam_amd.batch import pack_clouds
    fs, so = pack_clouds(srcs, dtype=np.float32); ft, to = pack_clouds(tgts, dtype=np.float32)
    I = np.tile(np.eye(4), (len(srcs), 1, 1))
    monkeypatch.setenv("LSLAM_ICP_QUEUE_TIMEOUT_US", "0.01")
    with pytest.raises(LslamError, match="gave up"):
        icp_batch(fs, so, ft, to, I, prune=True, epsilon=1e-3)
    monkeypatch.delenv("LSLAM_ICP_QUEUE_TIMEOUT_US")
    good = icp_batch(fs, so, ft, to, I, prune=True, epsilon=1e-3)
    ref = icp_batch(fs, so, ft, to, I, prune=False, epsilon=1e-3)
    np.testing.assert_array_equal(good.iterations, ref.iterations)
    np.testing.assert_allclose(good.T, ref.T, atol=1e-9)


@pytest.mark.parametrize("incr", ["1", "0"])
def test_task_queue_certificates_on_clouds_made_of_near_ties(icp, prec, incr, monkeypatch):
    """(also the test of the contested list: far more sources without a margin than a problem may list, exact ties that must not be listed)
    certificates and running sums forced on for clouds in which EVERY source has four candidates the float32 search cannot tell apart
    (lattice + nanometre perturbations, as in test_float64_exact_mode_on_clouds_made_of_near_ties), a start off the solution and twenty
    forced iterations: a near tie decided on the float64 points gets its margin from the exact runner-up, an exact tie none — the
    correspondences of every iteration must still be the all-pairs sweep's (the last ones are compared, and T, error and iteration count
    depend on all of them)"""
    from lidar_slam_amd.batch import icp_batch
    monkeypatch.setenv("LSLAM_ICP_QUEUE_CERTIFY", "1")
    monkeypatch.setenv("LSLAM_ICP_QUEUE_INCR", incr)
    rng = np.random.default_rng(99)
    g = np.stack(np.meshgrid(np.arange(40), np.arange(40)), -1).reshape(-1, 2) * 0.5 + 7.0
    srcs, tgts, inits = [], [], []
    for scale, th, sh in ((1e-9, 0.0, 0.0), (1e-7, 0.004, 0.01), (1e-6, -0.01, 0.03), (1e-4, 0.02, -0.05)):
        tgts.append(g[rng.permutation(g.shape[0])] + rng.uniform(-1, 1, g.shape) * scale)
        srcs.append(g[rng.integers(0, g.shape[0], 1300)] + 0.25 + rng.uniform(-1, 1, (1300, 2)) * scale)
        T0 = np.eye(4); T0[:2, :2] = [[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]]; T0[:2, 3] = sh
        inits.append(T0)
    fs, so = prec.pack(srcs); ft, to = prec.pack(tgts)
    kw = dict(epsilon=0.0, stopping_thresh=0.0, max_iters=19, return_correspondences=True)
    a = icp_batch(fs, so, ft, to, np.stack(inits), prune=False, **kw)
    b = icp_batch(fs, so, ft, to, np.stack(inits), prune=True, **kw)
    np.testing.assert_array_equal(b.iterations, a.iterations)
    np.testing.assert_array_equal(b.correspondences, a.correspondences)
    np.testing.assert_allclose(b.T, a.T, atol=1e-9)
    np.testing.assert_allclose(b.error, a.error, rtol=1e-9, atol=1e-12)


def test_task_queue_kernel_is_bitwise_reproducible_and_batch_independent(icp, prec):
    """the same pairs solved twice, alone, and in the middle of another batch: every bit of T, error and iteration count equal — the
    grouping of the float64 partial sums depends on a problem's size only (what the sharded driver relies on)"""
    from lidar_slam_amd.batch import icp_batch
    rng = np.random.default_rng(5)
    srcs, tgts = [], []
    for n, m in [(1081, 1000), (2600, 2300), (777, 800), (4096, 4096), (300, 1081)]:
        s, t = _random_problem(rng, 2, n, m, spread=8.0)
        srcs.append(s); tgts.append(t)
    kw = dict(epsilon=1e-4, prune=True)
    def solve(idx):
        fs, so = prec.pack([srcs[i] for i in idx]); ft, to = prec.pack([tgts[i] for i in idx])
        return icp_batch(fs, so, ft, to, np.tile(np.eye(4), (len(idx), 1, 1)), **kw)
    whole, again = solve(range(5)), solve(range(5))
    assert np.array_equal(whole.T, again.T) and np.array_equal(whole.error, again.error)
    for i in range(5):
        alone = solve([i])
        assert np.array_equal(alone.T[0], whole.T[i]) and alone.error[0] == whole.error[i] and alone.iterations[0] == whole.iterations[i], i
    mixed = solve([3, 0, 0, 4, 1, 2, 2])
    for pos, i in enumerate([3, 0, 0, 4, 1, 2, 2]):
        assert np.array_equal(mixed.T[pos], whole.T[i]), (pos, i)
