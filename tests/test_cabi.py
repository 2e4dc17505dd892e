"""The C-ABI library loads and exports every symbol include/lslam.h declares (no GPU needed)."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import ROOT, golden, split

HEADER = os.path.join(ROOT, "include", "lslam.h")


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(lslam_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_expected_surface():
    names = declared_functions()
    for must in ("lslam_icp_batch", "lslam_icp_batch_dev", "lslam_icp_large_dev", "lslam_nn_dev",
                 "lslam_kabsch", "lslam_sq_error", "lslam_ogm_create", "lslam_ogm_update",
                 "lslam_ogm_scan_counts", "lslam_last_error", "lslam_device_count"):
        assert must in names


def test_library_exports_every_declared_symbol():
    from lidar_slam_amd import _lib
    handle = _lib.lib()
    for name in declared_functions():
        assert hasattr(handle, name), f"{name} declared in lslam.h but not exported"
        assert name in _lib.SIGNATURES, f"{name} has no ctypes signature in _lib.py"
    assert handle.lslam_abi_version() == 1
    assert handle.lslam_device_count() >= 0


def test_argument_validation_needs_no_gpu():
    """bad arguments are rejected before any HIP call, with the reference's exception types"""
    from lidar_slam_amd.batch import icp_batch
    pts = np.zeros((4, 2), dtype=np.float32)
    off = np.array([0, 4], dtype=np.int64)
    with pytest.raises(IndexError):                       # empty target: reference raises IndexError
        icp_batch(pts, off, np.zeros((0, 2), np.float32), np.array([0, 0], np.int64), np.eye(4)[None])
    bad = pts.copy(); bad[1, 0] = np.nan
    with pytest.raises(ValueError):                       # NaN: reference raises ValueError (KDTree)
        icp_batch(bad, off, pts, off, np.eye(4)[None])
    with pytest.raises(ValueError):
        icp_batch(np.zeros((4, 5), np.float32), off, pts, off, np.eye(4)[None])
    tilt = np.eye(4); tilt[2, 0] = 0.1                    # would lift planar points out of z = 0
    with pytest.raises(ValueError):
        icp_batch(pts + 1, off, pts + 1, off, tilt[None])


def test_argument_validation_of_the_float64_entry_points_needs_no_gpu():
    """lslam_icp_batch_f64 / lslam_icp_large_f64 / lslam_host_staging reject bad arguments before any HIP call, with the
    same status codes (hence the same Python exceptions) as the float32 entry points"""
    import ctypes as C
    from lidar_slam_amd import _lib
    from lidar_slam_amd._lib import lib, ptr
    from lidar_slam_amd.batch import icp_batch
    pts = np.zeros((4, 2), dtype=np.float64)
    off = np.array([0, 4], dtype=np.int64)
    with pytest.raises(IndexError):
        icp_batch(pts, off, np.zeros((0, 2)), np.array([0, 0], np.int64), np.eye(4)[None])
    bad = pts.copy(); bad[1, 0] = np.inf
    with pytest.raises(ValueError):
        icp_batch(bad, off, pts, off, np.eye(4)[None])
    tilt = np.eye(4); tilt[2, 3] = 0.5
    with pytest.raises(ValueError):
        icp_batch(pts + 1, off, pts + 1, off, tilt[None])
    T = np.empty((4, 4)); e = np.empty(1); it = np.empty(1, np.int32)
    h = lib()
    assert h.lslam_icp_large_f64(ptr(pts), 4, ptr(pts), 0, 2, ptr(np.eye(4)), 0.01, 1e-4, 10, 0, ptr(T), ptr(e), ptr(it), None, 0) == _lib.ERR_EMPTY_TARGET
    assert h.lslam_icp_large_f64(ptr(pts), 0, ptr(pts), 4, 2, ptr(np.eye(4)), 0.01, 1e-4, 10, 0, ptr(T), ptr(e), ptr(it), None, 0) == _lib.ERR_BAD_ARG
    assert h.lslam_icp_large_f64(ptr(bad), 4, ptr(pts), 4, 2, ptr(np.eye(4)), 0.01, 1e-4, 10, 0, ptr(T), ptr(e), ptr(it), None, 0) == _lib.ERR_NONFINITE
    assert h.lslam_icp_large_f64(ptr(pts), 4, ptr(pts), 4, 5, ptr(np.eye(4)), 0.01, 1e-4, 10, 0, ptr(T), ptr(e), ptr(it), None, 0) == _lib.ERR_BAD_ARG
    out = C.c_void_p()
    assert h.lslam_host_staging(2, 64, C.byref(out)) == _lib.ERR_BAD_ARG and b"slot" in h.lslam_last_error()
    assert h.lslam_host_staging(0, 64, None) == _lib.ERR_BAD_ARG


def test_no_silent_cpu_fallback_without_a_device():
    import lidar_slam_amd
    if lidar_slam_amd.device_count() > 0:
        pytest.skip("a GPU is present")
    from lidar_slam_amd.modules import icp
    with pytest.raises(lidar_slam_amd.LslamError):
        icp.run_icp(np.random.rand(8, 2), np.random.rand(8, 2))


def test_solver_header_on_host_matches_reference_kabsch():
    """lslam_solve.h (the code lane 0 runs on the GPU) compiled for the host vs golden get_transform."""
    so = os.path.join(ROOT, "lidar_slam_amd", "lib", "liblslam_hostcheck.so")
    h = ctypes.CDLL(so)
    h.lslam_hostcheck_step.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    g = golden("kabsch.npz")
    n_planar = int(g["n_planar"])
    for k, (p, q) in enumerate(zip(split(g["p"], g["off"]), split(g["q"], g["off"]))):
        m = np.zeros(23)
        m[0] = p.shape[0]; m[1:4] = p.sum(0); m[4:7] = q.sum(0); m[7:16] = (p.T @ q).ravel()
        for planar in ([True, False] if (k < n_planar or p.shape[0] == 1) else [False]):
            T = np.empty((4, 4))
            h.lslam_hostcheck_step(m.ctypes.data_as(ctypes.c_void_p), int(planar), T.ctypes.data_as(ctypes.c_void_p))
            np.testing.assert_allclose(T, g["T"][k], atol=1e-10, err_msg=f"case {k} planar={planar}")


def test_stop_rules_on_host():
    so = os.path.join(ROOT, "lidar_slam_amd", "lib", "liblslam_hostcheck.so")
    h = ctypes.CDLL(so)
    h.lslam_hostcheck_stop.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_double, ctypes.c_double, ctypes.c_int]
    def run(errs, eps, thr, mx):
        e = np.array(errs, dtype=np.float64)
        return h.lslam_hostcheck_stop(e.ctypes.data_as(ctypes.c_void_p), len(errs), eps, thr, mx)
    assert run([5, 4, 3, 2, 1, 0.5], 0.0, 0.0, 3) == 4           # max_iters + 1 iterations at most
    assert run([5, 4, 3.99995, 1], 0.01, 1e-4, 2000) == 3        # Δ rule needs a previous error
    assert run([0.001, 5], 0.01, 1e-4, 2000) == 1                # epsilon fires first
    assert run([5] * 60, 0.0, 0.0, 49) == 50                     # "50 forced iterations" (I2)


def test_pose_chain_reproduces_reference_outputs_without_gpu():
    """lslam_chain_se2 is host arithmetic: the reference's committed relative poses chain to its poses"""
    from lidar_slam_amd.modules.localization import chain_poses
    g = golden("se2_outputs20.npz")
    np.testing.assert_allclose(chain_poses(g["relative_poses_scan_matching"]), g["poses_scan_matching"], atol=1e-12)
    assert chain_poses(np.zeros((0, 3, 3))).tolist() == [[0.0, 0.0, 0.0]]


def test_header_constants_match_the_python_binding():
    """flag bits and status codes of include/lslam.h are the ones lidar_slam_amd/_lib.py passes and maps"""
    from lidar_slam_amd import _lib
    text = open(HEADER).read()
    defs = {k: int(v.rstrip("u"), 0) for k, v in re.findall(r"#define\s+(LSLAM_[A-Z0-9_]+)\s+\(?(-?\d+u?)\)?", text)}
    assert defs["LSLAM_ICP_NORMALIZE_ERROR"] == _lib.ICP_NORMALIZE_ERROR
    assert defs["LSLAM_ICP_PRUNE"] == _lib.ICP_PRUNE and defs["LSLAM_ICP_ALL_PAIRS"] == _lib.ICP_ALL_PAIRS
    assert _lib.search_flags(None) == 0 and _lib.search_flags(True) == defs["LSLAM_ICP_PRUNE"] and _lib.search_flags(False) == defs["LSLAM_ICP_ALL_PAIRS"]
    for name, val in (("LSLAM_PRE_POLAR", _lib.PRE_POLAR), ("LSLAM_PRE_DBSCAN", _lib.PRE_DBSCAN), ("LSLAM_PRE_RADIAL", _lib.PRE_RADIAL)):
        assert defs[name] == val
    for name, val in (("LSLAM_OK", _lib.OK), ("LSLAM_ERR_BAD_ARG", _lib.ERR_BAD_ARG), ("LSLAM_ERR_EMPTY_TARGET", _lib.ERR_EMPTY_TARGET),
                      ("LSLAM_ERR_NONFINITE", _lib.ERR_NONFINITE), ("LSLAM_ERR_HIP", _lib.ERR_HIP)):
        assert defs[name] == val
