"""The CPU oracle replayed against every ICP golden vector captured from the reference."""
import numpy as np
import pytest

from conftest import golden, split
from oracle import icp_oracle as O


def _check(r, T, err, corr, iters, atol=1e-9):
    assert r.iterations == iters
    np.testing.assert_allclose(r.T, T, rtol=0, atol=atol)
    np.testing.assert_allclose(r.error, err, rtol=1e-9, atol=1e-12)
    np.testing.assert_array_equal(r.correspondences, corr)


@pytest.mark.parametrize("nn", ["kdtree", "brute"])
def test_planar_pairs(nn):
    g = golden("icp_planar.npz")
    srcs, corrs = split(g["src"], g["src_off"]), split(g["corr"], g["src_off"])
    for k, src in enumerate(srcs):
        r = O.run_icp_full(src, g["tgt"], g["init"][k], epsilon=float(g["epsilon"]), nn=nn)
        _check(r, g["T"][k], g["error"][k], corrs[k], g["iterations"][k])


def test_planar_pairs_device_recipe_stays_inside_tolerance():
    """fp32 points + fp32 argmin + fp64 moments (the HIP recipe) vs the float64 reference."""
    from conftest import TOL_ROT, TOL_TRANS
    g = golden("icp_planar.npz")
    for k, src in enumerate(split(g["src"], g["src_off"])):
        r = O.run_icp_full(src, g["tgt"], g["init"][k], epsilon=float(g["epsilon"]), recipe="device")
        assert r.iterations == g["iterations"][k]
        assert np.abs(r.T[:3, 3] - g["T"][k][:3, 3]).max() < TOL_TRANS
        assert np.abs(r.T[:3, :3] - g["T"][k][:3, :3]).max() < TOL_ROT


def test_forced_50_iterations():
    g = golden("icp_forced50.npz")
    srcs, corrs = split(g["src"], g["src_off"]), split(g["corr"], g["src_off"])
    for k, src in enumerate(srcs):
        r = O.run_icp_full(src, g["tgt"], g["init"][k], epsilon=0.0, stopping_thresh=0.0, max_iters=49)
        _check(r, g["T"][k], g["error"][k], corrs[k], 50)


def test_loop_closure_style_normalized_error():
    g = golden("icp_loopclose.npz")
    srcs, corrs = split(g["src"], g["src_off"]), split(g["corr"], g["src_off"])
    for k, src in enumerate(srcs):
        r = O.run_icp_full(src, g["tgt"], None, epsilon=float(g["epsilon"][k]), normalize_error=True)
        _check(r, g["T"][k], g["error"][k], corrs[k], g["iterations"][k])


@pytest.mark.parametrize("nn", ["kdtree", "brute"])
def test_3d_multistart(nn):
    g = golden("icp_3d.npz")
    for k in range(g["init"].shape[0]):
        r = O.run_icp_full(g["src"], g["tgt"], g["init"][k], epsilon=0.001, normalize_error=True, nn=nn)
        _check(r, g["T"][k], g["error"][k], g["corr"][k], g["iterations"][k])


def test_24_yaw_multistart_best_of():
    """the registration loop of code/icp_warm_up/test_icp.py:66-92 restated with the oracle"""
    g, m = golden("icp_3d.npz"), golden("icp_multistart.npz")
    src, tgt = g["src"], g["tgt"]
    errs = []
    for k, yaw in enumerate(np.linspace(0, 2 * np.pi, 24, endpoint=False)):
        T0 = np.eye(4)
        T0[:3, :3] = [[np.cos(yaw), -np.sin(yaw), 0], [np.sin(yaw), np.cos(yaw), 0], [0, 0, 1]]
        T0[:3, 3] = tgt.mean(0) - T0[:3, :3] @ src.mean(0)
        r = O.run_icp_full(src, tgt, T0, epsilon=0.001, normalize_error=True)
        np.testing.assert_allclose(r.T, m["T"][k], atol=1e-9)
        assert r.iterations == m["iterations"][k]
        errs.append(r.error)
    np.testing.assert_allclose(errs, m["error"], rtol=1e-9)
    assert int(np.argmin(errs)) == int(m["best"])


def test_kabsch_cases_svd_and_closed_forms():
    g = golden("kabsch.npz")
    ps, qs = split(g["p"], g["off"]), split(g["q"], g["off"])
    n_planar = int(g["n_planar"])
    reflections = 0
    for k, (p, q) in enumerate(zip(ps, qs)):
        T = g["T"][k]
        np.testing.assert_allclose(O.kabsch_svd(p, q), T, atol=1e-12)
        planar = k < n_planar or p.shape[0] == 1
        Tm = O.kabsch_from_moments(p.shape[0], p.sum(0), q.sum(0), p.T @ q, planar=planar)
        np.testing.assert_allclose(Tm, T, atol=1e-10)
        reflections += int(planar and T[2, 2] < 0)
    assert reflections >= 4          # the reflection branch (I5) is really exercised


def test_return_convention_and_errors():
    g = golden("icp_planar.npz")
    src, tgt = g["src"][: g["src_off"][1]], g["tgt"]
    T = O.run_icp(src, tgt, g["init"][0], epsilon=0.001)
    assert T.shape == (4, 4)
    T2, e = O.run_icp(src, tgt, g["init"][0], epsilon=0.001, return_error=True)
    T3, c = O.run_icp(src, tgt, g["init"][0], epsilon=0.001, return_correspondences=True)
    T4, e4, c4 = O.run_icp(src, tgt, g["init"][0], epsilon=0.001, return_error=True, return_correspondences=True)
    assert np.array_equal(T, T2) and np.array_equal(T, T3) and np.array_equal(T, T4)
    assert e == e4 and np.array_equal(c, c4) and c.dtype == np.int64
    with pytest.raises(IndexError):
        O.run_icp(src, np.zeros((0, 2)))
    bad = src.copy(); bad[3, 0] = np.nan
    with pytest.raises(ValueError):
        O.run_icp(bad, tgt)


def test_bruteforce_matches_kdtree_on_random_clouds():
    from scipy.spatial import KDTree
    rng = np.random.default_rng(0)
    for d in (2, 3):
        p = np.zeros((700, 3)); q = np.zeros((900, 3))
        p[:, :d] = rng.uniform(-10, 10, (700, d)); q[:, :d] = rng.uniform(-10, 10, (900, d))
        np.testing.assert_array_equal(O.nn_bruteforce(p, q), KDTree(q).query(p)[1])


def test_voxel_downsample():
    g = golden("voxel.npz")
    out = O.voxel_downsample(g["points"], float(g["voxel_size"]))
    np.testing.assert_allclose(out, g["out"], atol=1e-14)
