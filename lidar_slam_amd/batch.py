"""Batched ICP over the C ABI: host-array and device-resident entry points."""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

from . import _lib
from ._lib import check, lib, ptr


@dataclass
class IcpBatchResult:
    T: np.ndarray            # (S,4,4) float64
    error: np.ndarray        # (S,) float64
    iterations: np.ndarray   # (S,) int32
    correspondences: np.ndarray | None = None   # (total_src,) int32, CSR like src_off


class PackedScans(list):
    """A list of (n_i, d) arrays that are consecutive row slices of ONE packed array (``flat``,
    ``off``).  Behaves exactly like the plain list the reference's pre-processing returns; the
    batched entry points recognise it and skip re-packing 5 000 small arrays in the interpreter.

    The packed form is only a cache of the list: any mutation of the list (item assignment, ``sort``,
    ``reverse``, ``append`` …) drops it, so an edited list is honoured like a plain one (re-packed).
    Writing INTO an element (``z[i][:] = …``) writes into ``flat``: the elements are views."""

    def __init__(self, flat, off):
        super().__init__(flat[off[k]:off[k + 1]] for k in range(off.shape[0] - 1))
        self.flat = flat
        self.off = off

    def packed(self):
        """(flat, off) while the list still is the row slices of ``flat`` it was built from, else None"""
        if self.flat is None or len(self) != self.off.shape[0] - 1:
            return None
        return self.flat, self.off

    def _drop(self):
        self.flat = None
        self.off = None


def _invalidating(name):
    plain = getattr(list, name)

    def method(self, *a, **k):
        self._drop()
        return plain(self, *a, **k)
    method.__name__ = name
    return method


for _name in ("__setitem__", "__delitem__", "__iadd__", "__imul__", "append", "extend", "insert", "pop", "remove",
              "clear", "sort", "reverse"):
    setattr(PackedScans, _name, _invalidating(_name))


_STAGED_POINTS = {}          # (slot, dim, dtype) -> points the thread's pinned staging slot is known to hold


def release_thread_resources():
    """Hands the calling thread's library-owned device and pinned memory back (``lslam_release_thread_resources``): scratch
    arena, pinned staging buffers, the ordered copies of the pruned kernels, single-pair workspaces.  The next call
    allocates again.  Arrays previously returned by the staging helpers must not be used afterwards."""
    _STAGED_POINTS.clear()
    check(lib().lslam_release_thread_resources())


class StagedClouds(np.ndarray):
    """a packed cloud array living in the library's pinned staging buffer; ``.checked`` = every coordinate was seen finite
    while packing (the library then skips its own pass: LSLAM_ICP_INPUT_CHECKED)"""
    checked = False


def _pack_staged(clouds, dtype, slot):
    """list of arrays -> pinned staging buffer, in C with a few threads (csrc/pyhost); None when the helper cannot take
    this list (not built, an element that is not a C-contiguous float array, …)"""
    import os
    ph = _lib.pyhost()
    if ph is None or not isinstance(clouds, list) or not len(clouds) or isinstance(clouds, PackedScans):
        return None
    first = clouds[0]
    if not isinstance(first, np.ndarray) or first.ndim != 2 or first.shape[1] not in (2, 3):
        return None
    d = first.shape[1]
    want = dtype or (np.float32 if first.dtype == np.float32 else np.float64)
    off = np.empty(len(clouds) + 1, dtype=np.int64)
    threads = min(16, os.cpu_count() or 1)
    for _ in range(3):
        key = (slot, d, np.dtype(want).name)
        cap = _STAGED_POINTS.get(key, 0)
        try:
            arr = _lib.staging_array(slot, max(cap, 1) * d, want)
        except _lib.LslamError:                                      # no device / no pinned memory: NumPy packs
            return None
        r = ph.pack_list(clouds, arr.ctypes.data, cap, d, int(want == np.float64), off.ctypes.data, threads)
        if r is None:
            return None
        total, finite, n_f64 = r
        if dtype is None and want == np.float32 and n_f64:          # a float64 scan among float32 ones: keep the precision
            want = np.float64
            continue
        if finite == -1:                                             # first time at this size: grow the slot and pack again
            _STAGED_POINTS.clear()                                   # (the slot is shared by all shapes: forget what it held)
            _STAGED_POINTS[key] = total + total // 8 + 1024
            continue
        if finite == 0:                                             # what _lib.check raises for LSLAM_ERR_NONFINITE (the reference: ValueError)
            raise ValueError("point clouds must be finite (NaN or Inf found)")
        flat = arr[: total * d].reshape(total, d).view(StagedClouds)
        flat.checked = True
        return flat, off
    return None


def pack_clouds(clouds, dtype=None, staging=None):
    """list of (n_i, d) arrays -> (``dtype`` (Σn_i, d) C-contiguous, int64 offsets (len+1,)).
    ``dtype=None`` keeps the points as they are handed over: float32 if EVERY cloud is float32 (``icp_batch`` then searches
    them as they are), float64 otherwise — the reference's own type; ``icp_batch`` then runs in float64-exact mode.
    An explicit float32 rounds float64 clouds in the same pass (half the upload, float32-exact search).
    ``staging=0|1``: pack into that pinned staging slot of the library with the C helper (several threads, finiteness
    checked on the way, upload at PCIe speed) — the result is only valid until the slot is used again, so this is for
    callers that hand it to the library at once."""
    if staging is not None:
        staged = _pack_staged(clouds, dtype, staging)
        if staged is not None:
            return staged
    if dtype is None and not isinstance(clouds, PackedScans):
        clouds = [c if isinstance(c, np.ndarray) else np.asarray(c) for c in clouds]
    if dtype is None:
        if isinstance(clouds, PackedScans) and clouds.packed() is not None:
            dtype = np.float32 if clouds.packed()[0].dtype == np.float32 else np.float64
        else:
            dtype = np.float32 if (len(clouds) and all(getattr(c, "dtype", None) == np.float32 for c in clouds)) else np.float64
    if isinstance(clouds, PackedScans):
        packed = clouds.packed()
        if packed is not None:
            return np.ascontiguousarray(packed[0], dtype=dtype), packed[1]
    if not isinstance(clouds, PackedScans):                            # elements may be Python lists or other array-likes
        clouds = [c if isinstance(c, np.ndarray) else np.asarray(c) for c in clouds]
    d = clouds[0].shape[1] if len(clouds) and clouds[0].ndim == 2 else 2
    off = np.zeros(len(clouds) + 1, dtype=np.int64)
    off[1:] = np.cumsum([c.shape[0] for c in clouds])
    if any(c.ndim != 2 or c.shape[1] != d for c in clouds):
        raise ValueError("all clouds of a batch must be (n_i, d) with the same d")
    if not len(clouds):
        return np.zeros((0, d), dtype), off
    flat = np.empty((int(off[-1]), d), dtype=dtype)
    np.concatenate(clouds, axis=0, out=flat, casting="same_kind")      # one pass: gather (and round, for float32)
    return flat, off


def icp_batch(src, src_off, tgt, tgt_off, init_T, epsilon=0.01, max_iters=2000, stopping_thresh=0.0001,
              normalize_error=False, return_correspondences=False, device=0, prune=None) -> IcpBatchResult:
    """S independent run_icp problems (modules/icp.py:123-189) in one launch, host arrays in/out.

    src, tgt: (N,d) packed clouds (d = 2 or 3; they may be the same array for a scan sequence), both float64 or
    both float32; src_off, tgt_off: int64 (S+1,) point offsets; init_T: (S,4,4) float64.
    float64 clouds run in FLOAT64-EXACT mode (``lslam_icp_batch_f64``): the float32 search only proposes candidates,
    near ties are decided and the moments accumulated on the float64 points — correspondences are those of the
    reference's float64 search (modules/icp.py:29-41), T / error agree with it to summation order.  float32 clouds
    (``lslam_icp_batch``) are searched as they are: exact float32 nearest neighbours.
    ``prune``: None = the library picks the search strategy (pruning for planar batches from 256 points per cloud and 3-D
    batches from 256), True = exact spatial pruning (planar problems of ≤ 4096 points per cloud, 3-D of ≤ 4096 target /
    8191 source points: both clouds sorted spatially, target chunks that provably cannot hold a nearer point skipped),
    False = the all-pairs sweep.  Same correspondences and
    iteration counts either way."""
    same = tgt is src
    checked = bool(getattr(src, "checked", False)) and bool(getattr(tgt, "checked", False))     # packed by the C helper: finiteness already seen
    src = np.asarray(src)
    dtype = np.float32 if (src.dtype == np.float32 and np.asarray(tgt).dtype == np.float32) else np.float64
    src = np.ascontiguousarray(src, dtype=dtype)
    tgt = src if same else np.ascontiguousarray(tgt, dtype=dtype)
    if src.ndim != 2 or src.shape[1] not in (2, 3) or tgt.ndim != 2 or tgt.shape[1] != src.shape[1]:
        raise ValueError("clouds must be (N,2) or (N,3) with equal dimension")
    src_off = np.ascontiguousarray(src_off, dtype=np.int64)
    tgt_off = np.ascontiguousarray(tgt_off, dtype=np.int64)
    S = src_off.shape[0] - 1
    if tgt_off.shape[0] - 1 != S:
        raise ValueError("src_off and tgt_off must describe the same number of problems")
    if S and (src_off[-1] > src.shape[0] or tgt_off[-1] > tgt.shape[0]):
        raise ValueError("offsets run past the end of the point buffers")
    init_T = np.ascontiguousarray(init_T, dtype=np.float64).reshape(S, 4, 4)
    out_T = np.empty((S, 4, 4), dtype=np.float64)
    out_err = np.empty(S, dtype=np.float64)
    out_it = np.empty(S, dtype=np.int32)
    corr = np.empty(int(src_off[-1]) if S else 0, dtype=np.int32) if return_correspondences else None
    flags = (_lib.ICP_NORMALIZE_ERROR if normalize_error else 0) | _lib.search_flags(prune) | (_lib.ICP_INPUT_CHECKED if checked else 0)
    entry = lib().lslam_icp_batch_f64 if dtype == np.float64 else lib().lslam_icp_batch
    check(entry(ptr(src), ptr(src_off), ptr(tgt), ptr(tgt_off), src.shape[1], ptr(init_T), S,
                float(epsilon), float(stopping_thresh), int(max_iters), flags,
                ptr(out_T), ptr(out_err), ptr(out_it), ptr(corr), int(device)))
    return IcpBatchResult(out_T, out_err, out_it, corr)


def icp_batch_dev(d_src, d_src_off, d_tgt, d_tgt_off, dim, d_init_T, S, max_src_n, max_tgt_n,
                  d_out_T, d_out_err, d_out_iters, d_out_corr=None, epsilon=0.01, max_iters=2000,
                  stopping_thresh=0.0001, normalize_error=False, stream=0, prune=None, d_src64=None, d_tgt64=None) -> None:
    """Same on DEVICE pointers (ints, e.g. torch.Tensor.data_ptr()); enqueues and returns.  ``d_src64`` / ``d_tgt64``:
    the same clouds in float64 (laid out alike) → float64-exact mode (``lslam_icp_batch_dev64``)."""
    flags = (_lib.ICP_NORMALIZE_ERROR if normalize_error else 0) | _lib.search_flags(prune)
    check(lib().lslam_icp_batch_dev64(d_src, d_src64 or None, d_src_off, d_tgt, d_tgt64 or None, d_tgt_off, int(dim), d_init_T, int(S),
                                      int(max_src_n), int(max_tgt_n), float(epsilon), float(stopping_thresh),
                                      int(max_iters), flags, d_out_T, d_out_err, d_out_iters,
                                      d_out_corr or None, stream or None))


def icp_large_dev(d_src, n, d_tgt, m, dim, init_T, epsilon=0.01, max_iters=2000, stopping_thresh=0.0001,
                  normalize_error=False, d_out_corr=None, poll_every=8, stream=0, prune=None, d_src64=None, d_tgt64=None):
    """run_icp on one large cloud pair resident on the device; returns (T, error, iterations).
    ``prune``: None = automatic, True = spatially ordered clouds (Hilbert curve) + a box hierarchy over the target, False = all pairs;
    exact results either way.  ``d_src64`` / ``d_tgt64``: the same clouds in float64 → float64-exact mode."""
    init_T = np.ascontiguousarray(init_T, dtype=np.float64).reshape(4, 4)
    out_T = np.empty((4, 4), dtype=np.float64)
    out_err = np.empty(1, dtype=np.float64)
    out_it = np.empty(1, dtype=np.int32)
    flags = (_lib.ICP_NORMALIZE_ERROR if normalize_error else 0) | _lib.search_flags(prune)
    check(lib().lslam_icp_large_dev64(d_src, d_src64 or None, int(n), d_tgt, d_tgt64 or None, int(m), int(dim), ptr(init_T), float(epsilon),
                                      float(stopping_thresh), int(max_iters), flags, int(poll_every),
                                      ptr(out_T), ptr(out_err), ptr(out_it), d_out_corr or None, stream or None))
    return out_T, float(out_err[0]), int(out_it[0])


def nn_dev(d_src, n, d_tgt, m, dim, T, d_out_idx, d_out_d2=None, stream=0) -> None:
    T = None if T is None else np.ascontiguousarray(T, dtype=np.float64).reshape(4, 4)
    check(lib().lslam_nn_dev(d_src, int(n), d_tgt, int(m), int(dim), ptr(T), d_out_idx, d_out_d2 or None,
                             stream or None))
