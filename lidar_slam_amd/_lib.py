"""ctypes binding of ``liblslam_hip.so`` (the C ABI of ``include/lslam.h``)."""
from __future__ import annotations

import ctypes as C
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("LSLAM_LIB") or os.path.join(_HERE, "lib", "liblslam_hip.so")      # LSLAM_LIB: A/B runs of two builds

OK, ERR_BAD_ARG, ERR_EMPTY_TARGET, ERR_NONFINITE, ERR_HIP = 0, -1, -2, -3, -4
ICP_NORMALIZE_ERROR = 1
ICP_PRUNE = 2
ICP_ALL_PAIRS = 4
ICP_INPUT_CHECKED = 8


def search_flags(prune):
    """prune=None: the library picks (LSLAM_ICP_* in include/lslam.h); True / False force pruning / the all-pairs sweep"""
    return 0 if prune is None else (ICP_PRUNE if prune else ICP_ALL_PAIRS)
PRE_POLAR, PRE_DBSCAN, PRE_RADIAL = 1, 2, 4


class LslamError(RuntimeError):
    """A HIP failure (no device, launch error, out of memory) reported by the library."""


_lib = None

_f32p = C.POINTER(C.c_float)
_f64p = C.POINTER(C.c_double)
_i64p = C.POINTER(C.c_int64)
_i32p = C.POINTER(C.c_int32)
_u8p = C.POINTER(C.c_uint8)

# name -> (restype, argtypes); mirrors include/lslam.h line by line
SIGNATURES = {
    "lslam_abi_version": (C.c_int, []),
    "lslam_last_error": (C.c_char_p, []),
    "lslam_device_count": (C.c_int, []),
    "lslam_set_device": (C.c_int, [C.c_int]),
    "lslam_valu_probe": (C.c_int, [C.c_int, C.c_void_p]),
    "lslam_host_staging": (C.c_int, [C.c_int, C.c_size_t, C.POINTER(C.c_void_p)]),
    "lslam_release_thread_resources": (C.c_int, []),
    "lslam_icp_large_starts": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_int32, C.c_double,
                                         C.c_double, C.c_int32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
    "lslam_icp_large_starts_f64": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_int32, C.c_double,
                                             C.c_double, C.c_int32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
    "lslam_icp_coop_fallbacks": (C.c_int, []),
    "lslam_icp_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                  C.c_int32, C.c_double, C.c_double, C.c_int32, C.c_uint32, C.c_void_p,
                                  C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
    "lslam_icp_batch_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                      C.c_int32, C.c_int64, C.c_int64, C.c_double, C.c_double, C.c_int32,
                                      C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_void_p]),
    "lslam_icp_batch_f64": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                      C.c_int32, C.c_double, C.c_double, C.c_int32, C.c_uint32, C.c_void_p,
                                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
    "lslam_icp_batch_dev64": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                        C.c_void_p, C.c_int32, C.c_int64, C.c_int64, C.c_double, C.c_double, C.c_int32,
                                        C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "lslam_icp_large_dev": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int, C.c_void_p,
                                      C.c_double, C.c_double, C.c_int32, C.c_uint32, C.c_int32, C.c_void_p,
                                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "lslam_icp_large_dev64": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p,
                                        C.c_double, C.c_double, C.c_int32, C.c_uint32, C.c_int32, C.c_void_p,
                                        C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "lslam_icp_large_f64": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_double,
                                      C.c_double, C.c_int32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.c_int]),
    "lslam_icp_large": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_double,
                                  C.c_double, C.c_int32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p,
                                  C.c_void_p, C.c_int]),
    "lslam_nn": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p,
                           C.c_void_p, C.c_int]),
    "lslam_nn_f64": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_int]),
    "lslam_nn_dev": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int, C.c_void_p,
                               C.c_void_p, C.c_void_p, C.c_void_p]),
    "lslam_kabsch": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int]),
    "lslam_sq_error": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_int]),
    "lslam_voxel_downsample": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_double, C.c_void_p, C.c_void_p, C.c_int]),
    "lslam_chain_se2": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p]),
    "lslam_lidar_preprocess": (C.c_int, [C.c_uint32, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_double,
                                         C.c_double, C.c_double, C.c_double, C.c_void_p, C.c_void_p, C.c_double,
                                         C.c_int32, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
    "lslam_scans_create": (C.c_int, [C.c_uint32, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_double,
                                     C.c_double, C.c_double, C.c_double, C.c_void_p, C.c_void_p, C.c_double, C.c_int32,
                                     C.c_double, C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]),
    "lslam_scans_destroy": (C.c_int, [C.c_void_p]),
    "lslam_scans_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "lslam_scans_download": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "lslam_scans_match_sequence": (C.c_int, [C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_int32, C.c_uint32,
                                             C.c_void_p, C.c_void_p, C.c_void_p]),
    "lslam_ogm_update_scans": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "lslam_ogm_create": (C.c_int, [C.c_double, C.c_double, C.c_double, C.c_int32, C.c_int32, C.c_double,
                                   C.c_double, C.c_int, C.POINTER(C.c_void_p)]),
    "lslam_ogm_destroy": (C.c_int, [C.c_void_p]),
    "lslam_ogm_set_params": (C.c_int, [C.c_void_p, C.c_double, C.c_double]),
    "lslam_ogm_update": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32]),
    "lslam_ogm_scan_counts": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]),
    "lslam_ogm_read_logodds": (C.c_int, [C.c_void_p, C.c_void_p]),
    "lslam_ogm_write_logodds": (C.c_int, [C.c_void_p, C.c_void_p]),
    "lslam_ogm_binarize": (C.c_int, [C.c_void_p, C.c_void_p]),
}


def lib():
    """Load the shared library once.  Raises (never falls back) when it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise LslamError(
            f"{LIB_PATH} is missing: build it with `make -C lidar_slam_amd/csrc` "
            "(or `python -c 'import __graft_entry__ as g; g.build()'`). There is no CPU fallback.")
    # When torch is (or will be) in the process, its bundled libamdhip64 must be the one HIP
    # runtime both sides use: it has the same SONAME, so loading torch first makes ours bind to it.
    if "torch" not in sys.modules and os.environ.get("LSLAM_NO_TORCH") != "1":
        try:
            import torch  # noqa: F401
        except Exception:
            pass
    handle = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(handle, name)
        fn.restype = res
        fn.argtypes = args
    _lib = handle
    return _lib


def csrc_hash() -> str:
    """sha256 over the library's sources (csrc/*.hip, csrc/*.h, include/lslam.h, the Makefile) in name order: identifies
    the kernels a profile was taken on (tools/profile_round.sh stores it beside the counters; bench.py reports
    profile-derived figures only while it still matches)."""
    import glob
    import hashlib
    root = os.path.dirname(_HERE)
    files = sorted(glob.glob(os.path.join(_HERE, "csrc", "*.hip")) + glob.glob(os.path.join(_HERE, "csrc", "*.h")) +
                   [os.path.join(_HERE, "csrc", "Makefile"), os.path.join(root, "include", "lslam.h")])
    h = hashlib.sha256()
    for f in files:
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


_pyhost = None


def pyhost():
    """the CPython packing helper (csrc/pyhost/lslam_pyhost.cpp), or None when it is not built (NumPy packs then)"""
    global _pyhost
    if _pyhost is None:
        path = os.path.join(_HERE, "lib", "_lslam_pyhost.so")
        _pyhost = False
        if os.path.exists(path):
            import importlib.util
            spec = importlib.util.spec_from_file_location("_lslam_pyhost", path)
            try:
                mod = importlib.util.module_from_spec(spec)
                spec.loader.exec_module(mod)
                _pyhost = mod
            except Exception:
                _pyhost = False
    return _pyhost or None


def staging_array(slot, count, dtype):
    """numpy view of ``count`` elements of the thread's pinned staging buffer ``slot`` (valid until the slot is asked for again)"""
    import numpy as np
    nbytes = max(int(count), 1) * np.dtype(dtype).itemsize
    p = C.c_void_p()
    check(lib().lslam_host_staging(int(slot), nbytes, C.byref(p)))
    return np.frombuffer((C.c_char * nbytes).from_address(p.value), dtype=dtype, count=int(count))


def last_error() -> str:
    return lib().lslam_last_error().decode("utf-8", "replace")


def check(rc: int) -> None:
    """Map a C status to the exception the reference raises for the same input."""
    if rc == OK:
        return
    msg = last_error()
    if rc == ERR_EMPTY_TARGET:
        raise IndexError(msg)                    # reference: pc2[correspondences] on an empty cloud
    if rc == ERR_NONFINITE:
        raise ValueError(msg)                    # reference: KDTree "'x' must be finite"
    if rc == ERR_BAD_ARG:
        raise ValueError(msg)
    raise LslamError(msg or f"lslam error {rc}")


def device_count() -> int:
    return int(lib().lslam_device_count())


def ptr(a):
    """ctypes void* of a C-contiguous numpy array (None -> NULL)."""
    if a is None:
        return None
    return a.ctypes.data_as(C.c_void_p)
