"""ctypes access to oracle/csrc/oracle_icp.c (TEST INFRASTRUCTURE — see oracle/__init__.py)."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle_icp.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            subprocess.run(["make", "-C", os.path.join(_HERE, "csrc")], check=True)
        _lib = C.CDLL(_SO)
        _lib.oracle_run_icp.restype = C.c_int
        _lib.oracle_run_icp.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_double,
                                        C.c_int, C.c_double, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                        C.c_void_p, C.c_void_p]
        _lib.oracle_nn_f32.restype = None
        _lib.oracle_nn_f32.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p]
        _lib.oracle_kabsch.restype = None
        _lib.oracle_kabsch.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _xyz(pc):
    pc = np.asarray(pc, dtype=np.float64)
    out = np.zeros((pc.shape[0], 3))
    out[:, : pc.shape[1]] = pc
    return np.ascontiguousarray(out)


def run_icp(pc1, pc2, init_transform=None, epsilon=0.01, max_iters=2000, stopping_thresh=0.0001,
            normalize_error=False, recipe="f64"):
    """-> (T, error, correspondences int64, iterations)"""
    planar = np.asarray(pc1).shape[1] == 2 and np.asarray(pc2).shape[1] == 2
    src, tgt = _xyz(pc1), _xyz(pc2)
    if tgt.shape[0] == 0:
        raise IndexError("empty target cloud")
    T0 = np.ascontiguousarray(np.eye(4) if init_transform is None else init_transform, dtype=np.float64)
    T = np.empty((4, 4)); err = np.empty(1); corr = np.empty(max(src.shape[0], 1), dtype=np.int32)
    it = lib().oracle_run_icp(_p(src), src.shape[0], _p(tgt), tgt.shape[0], _p(T0), float(epsilon),
                              int(max_iters), float(stopping_thresh), int(normalize_error),
                              1 if recipe == "device" else 0, 2 if planar else 3, _p(T), _p(err), _p(corr))
    if it < 0:
        raise MemoryError
    return T, float(err[0]), corr[: src.shape[0]].astype(np.int64), it


def nn_f32(p32, q32):
    """exact float32 (fmaf) brute-force NN -> (idx int32, d2 float32)"""
    p32 = np.ascontiguousarray(p32, dtype=np.float32)
    q32 = np.ascontiguousarray(q32, dtype=np.float32)
    idx = np.empty(p32.shape[0], dtype=np.int32)
    d2 = np.empty(p32.shape[0], dtype=np.float32)
    lib().oracle_nn_f32(_p(p32), p32.shape[0], _p(q32), q32.shape[0], p32.shape[1], _p(idx), _p(d2))
    return idx, d2


def kabsch(p, q):
    p, q = _xyz(p), _xyz(q)
    T = np.empty((4, 4))
    lib().oracle_kabsch(_p(p), _p(q), p.shape[0], _p(T))
    return T
