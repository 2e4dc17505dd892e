"""The host-side native code under AddressSanitizer + UndefinedBehaviorSanitizer (and the packing helper's threads under
ThreadSanitizer): `make -C lidar_slam_amd/csrc asan tsan` builds instrumented copies of the solver header's host build
(csrc/hostcheck), the CPython packing helper (csrc/pyhost) and the C oracle (oracle/csrc); a child python3 with the sanitizer
runtime preloaded drives them with ragged / empty / non-contiguous / non-finite scan lists and degenerate moments.  CPU only —
the GPU pool has no device sanitizer (race detection / sanitizers)."""
import os
import subprocess
import sys
import textwrap

import pytest

from conftest import ROOT

CSRC = os.path.join(ROOT, "lidar_slam_amd", "csrc")
SAN = os.path.join(ROOT, "lidar_slam_amd", "lib_san")


def _runtime(name):
    path = subprocess.run(["gcc", f"-print-file-name={name}"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(path) or not os.path.exists(path):
        pytest.skip(f"{name} is not installed beside gcc")
    return path


@pytest.fixture(scope="module")
def built():
    subprocess.run(["make", "-C", CSRC, "asan", "tsan"], check=True, capture_output=True)
    return SAN


DRIVER = textwrap.dedent("""
    import ctypes as C, importlib.util, os, sys
    import numpy as np
    san = sys.argv[1]
    spec = importlib.util.spec_from_file_location("_lslam_pyhost", os.path.join(san, "_lslam_pyhost.so"))
    ph = importlib.util.module_from_spec(spec); spec.loader.exec_module(ph)
    rng = np.random.default_rng(0)

    def pack(clouds, dim, f64, threads, cap=None):
        total = sum(int(np.asarray(c).shape[0]) for c in clouds if hasattr(c, "shape") and np.asarray(c).ndim == 2)
        cap = total if cap is None else cap
        dst = np.full((max(cap, 1), dim), 7.0, dtype=np.float64 if f64 else np.float32)
        off = np.zeros(len(clouds) + 1, dtype=np.int64)
        r = ph.pack_list(clouds, dst.ctypes.data, cap, dim, int(f64), off.ctypes.data, threads)
        return r, dst, off

    # ragged list with empty scans, both dtypes mixed, enough points for several threads
    clouds = [rng.normal(size=(n, 2)).astype(np.float32 if k % 3 else np.float64) for k, n in enumerate([0, 5, 1081, 0, 700, 1, 60000, 3, 0, 40000])]
    for f64 in (0, 1):
        for threads in (1, 7, 64):
            r, dst, off = pack(clouds, 2, f64, threads)
            assert r is not None and r[0] == sum(c.shape[0] for c in clouds) and r[1] == 1, r
            ref = np.concatenate([c.astype(dst.dtype) for c in clouds])
            assert np.array_equal(dst[: r[0]], ref) and off[-1] == r[0]
    # destination too small: nothing is written
    r, dst, off = pack(clouds, 2, 1, 4, cap=10)
    assert r[1] == -1 and np.all(dst == 7.0)
    # declined inputs: non-contiguous, wrong dimension, integer dtype, a Python list element, not a list at all
    base = rng.normal(size=(50, 4))
    for bad in ([base[:, :2]], [np.zeros((4, 3))], [np.zeros((4, 2), dtype=np.int32)], [[[0.0, 1.0]]], [np.zeros(6)]):
        assert pack(bad, 2, 1, 2)[0] is None
    assert ph.pack_list((np.zeros((2, 2)),), 0, 0, 2, 1, 0, 1) is None
    # non-finite values are reported, also when only the CONVERSION overflows (a finite double beyond the float32 range)
    nf = [np.array([[0.0, 1.0], [np.nan, 2.0]]), np.ones((3, 2))]
    assert pack(nf, 2, 1, 2)[0][1] == 0
    big = [np.array([[1e300, 0.0]]), np.ones((2, 2))]
    assert pack(big, 2, 1, 1)[0][1] == 1 and pack(big, 2, 0, 1)[0][1] == 0
    # 3-D scans
    r, dst, off = pack([rng.normal(size=(n, 3)) for n in (10, 0, 33)], 3, 0, 3)
    assert r[0] == 43 and r[1] == 1

    if len(sys.argv) > 2 and sys.argv[2] == "asan":
        # the solver header on degenerate and hostile moments: n = 0 and 1, zero / rank-one / huge / NaN cross-covariances
        h = C.CDLL(os.path.join(san, "liblslam_hostcheck.so"))
        h.lslam_hostcheck_step.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        h.lslam_hostcheck_error.restype = C.c_double; h.lslam_hostcheck_error.argtypes = [C.c_void_p, C.c_int]
        h.lslam_hostcheck_stop.restype = C.c_int; h.lslam_hostcheck_stop.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_int]
        T = np.empty(16)
        cases = [np.zeros(23)]
        one = np.zeros(23); one[0] = 1; one[1:4] = [1, 2, 3]; one[4:7] = [2, 3, 4]; one[7:16] = np.outer([1, 2, 3], [2, 3, 4]).ravel(); cases.append(one)
        for scale in (1e-300, 1.0, 1e150):
            m = rng.normal(size=23) * scale; m[0] = 50; cases.append(m)
        r1 = np.zeros(23); r1[0] = 10; r1[7:16] = np.outer([1, 0, 0], [0, 1, 0]).ravel() * 5; cases.append(r1)
        bad = rng.normal(size=23); bad[0] = 10; bad[9] = np.nan; cases.append(bad)
        inf = rng.normal(size=23); inf[0] = 10; inf[7] = np.inf; cases.append(inf)
        for m in cases:
            m = np.ascontiguousarray(m)
            for planar in (0, 1):
                h.lslam_hostcheck_step(m.ctypes.data, planar, T.ctypes.data)
            for norm in (0, 1):
                h.lslam_hostcheck_error(m.ctypes.data, norm)
        errs = np.array([5.0, 4.0, np.nan, 3.0, 3.0, 0.0])
        assert h.lslam_hostcheck_stop(errs.ctypes.data, 6, 0.01, 1e-4, 3) >= 1
        # the C oracle: ragged tiny problems, duplicates, a single point, both recipes
        o = C.CDLL(os.path.join(san, "liboracle_icp.so"))
        o.oracle_run_icp.restype = C.c_int
        o.oracle_run_icp.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_double, C.c_int, C.c_double, C.c_int, C.c_int, C.c_int,
                                     C.c_void_p, C.c_void_p, C.c_void_p]
        for n, m_ in ((1, 1), (2, 5), (37, 3), (200, 200), (5, 64)):
            src = np.zeros((n, 3)); src[:, :2] = rng.normal(size=(n, 2))
            tgt = np.zeros((m_, 3)); tgt[:, :2] = rng.normal(size=(m_, 2)); tgt[m_ // 2] = tgt[0]
            T0 = np.eye(4); To = np.empty((4, 4)); e = np.empty(1); corr = np.empty(max(n, 1), dtype=np.int32)
            for recipe in (0, 1):
                it = o.oracle_run_icp(src.ctypes.data, n, tgt.ctypes.data, m_, T0.ctypes.data, 1e-3, 30, 1e-4, 0, recipe, 2, To.ctypes.data, e.ctypes.data, corr.ctypes.data)
                assert it >= 1 and corr[:n].min() >= 0 and corr[:n].max() < m_
    print("sanitizer driver: ok")
""")


def _drive(runtime, san_dir, mode, extra_env):
    env = dict(os.environ, LD_PRELOAD=runtime, **extra_env)
    r = subprocess.run([sys.executable, "-c", DRIVER, san_dir, mode], capture_output=True, text=True, env=env, timeout=600)
    report = r.stdout[-3000:] + r.stderr[-6000:]
    assert r.returncode == 0 and "sanitizer driver: ok" in r.stdout, report
    for marker in ("ERROR: AddressSanitizer", "runtime error:", "WARNING: ThreadSanitizer", "ERROR: ThreadSanitizer"):
        assert marker not in report, report


def test_host_code_under_asan_and_ubsan(built):
    _drive(_runtime("libasan.so"), os.path.join(built, "asan"), "asan",
           {"ASAN_OPTIONS": "detect_leaks=0:abort_on_error=0:halt_on_error=1", "UBSAN_OPTIONS": "print_stacktrace=1:halt_on_error=1"})


def test_packing_threads_under_tsan(built):
    _drive(_runtime("libtsan.so"), os.path.join(built, "tsan"), "tsan", {"TSAN_OPTIONS": "halt_on_error=1:report_signal_unsafe=0"})
