"""Multi-start registration of the reference's ICP drill (code/icp_warm_up/test_icp.py:66-92) on MI355X.

The reference aligns a canonical object model to each measured cloud by running ``run_icp`` from 24
yaw starts (rotation about z, translation matching the centroids) and keeping the start with the
lowest normalised error; clouds above 20 000 points are voxel-downsampled (5 mm) first.  Here the
starts of one target are ONE batched launch (one workgroup per start, clouds shared through the
same device buffers) when the clouds are small, and back-to-back whole-chip runs when they are
not; the best-of rule, the arguments of ``run_icp`` and the returned quantities are the reference's.
File loading and plotting of the reference script are out of scope.
"""
from __future__ import annotations

import numpy as np

from ._lib import ICP_NORMALIZE_ERROR, check, lib, ptr
from .batch import icp_batch
from .modules import icp

# above this many pair evaluations per iteration one start keeps the whole chip busy by itself
SEQUENTIAL_PAIR_EVALS = 2_000_000


def T_from_yaw(yaw):
    """4x4 rotation about z (test_icp.py:44-50)"""
    c, s = np.cos(yaw), np.sin(yaw)
    return np.array([[c, -s, 0.0, 0.0], [s, c, 0.0, 0.0], [0.0, 0.0, 1.0, 0.0], [0.0, 0.0, 0.0, 1.0]])


def yaw_starts(source_pc, target_pc, n_starts=24):
    """The initial transforms of test_icp.py:70-74: yaw k·2π/n about z, translation such that the rotated
    source centroid lands on the target centroid.  Returns (n_starts, 4, 4) float64."""
    src = np.asarray(source_pc, dtype=np.float64)
    tgt = np.asarray(target_pc, dtype=np.float64)
    a, b = np.mean(src, axis=0), np.mean(tgt, axis=0)
    T = np.stack([T_from_yaw(y) for y in np.linspace(0, 2 * np.pi, n_starts, endpoint=False)])
    T[:, 0:3, 3] = b[None, :] - np.einsum("kij,j->ki", T[:, 0:3, 0:3], a)
    return T


def _cloud_dtype(a, b):
    return np.float32 if (np.asarray(a).dtype == np.float32 and np.asarray(b).dtype == np.float32) else np.float64


def register_multi_start(source_pc, target_pc, n_starts=24, epsilon=0.001, voxel_size=0.005,
                         downsample_above=20000, max_iters=2000, stopping_thresh=0.0001,
                         return_all=False):
    """Best pose of ``n_starts`` yaw-started ICP runs (test_icp.py:66-92).

    Returns ``(best_pose (4,4) f64, best_error np.float64)`` — the first start reaching the lowest
    error wins, as the reference's ``err < best_error`` scan does — and with ``return_all`` also
    ``(poses (S,4,4), errors (S,), iterations (S,))`` of every start.
    """
    source_pc = np.asarray(source_pc)
    target_pc = np.asarray(target_pc)
    init = yaw_starts(source_pc, target_pc, n_starts)          # centroids of the FULL clouds (test_icp.py:72-74)
    if source_pc.shape[0] > downsample_above or target_pc.shape[0] > downsample_above:
        src, tgt = icp.voxel_downsample(source_pc, voxel_size), icp.voxel_downsample(target_pc, voxel_size)
    else:
        src, tgt = source_pc, target_pc
    n, m, S = src.shape[0], tgt.shape[0], init.shape[0]
    d = 2 if (src.shape[1] == 2 and tgt.shape[1] == 2) else 3
    # float64 clouds run in float64-exact mode (the reference's float64 search); float32 input is searched as it is
    dtype = _cloud_dtype(source_pc, target_pc)
    s32 = np.ascontiguousarray(icp._xyz(src)[:, :d], dtype=dtype)
    t32 = np.ascontiguousarray(icp._xyz(tgt)[:, :d], dtype=dtype)
    if n * m > SEQUENTIAL_PAIR_EVALS:
        # whole-chip runs, back to back on ONE upload of the pair (lslam_icp_large_starts)
        poses, errors, iters = np.empty((S, 4, 4)), np.empty(S), np.zeros(S, dtype=np.int32)
        entry = lib().lslam_icp_large_starts_f64 if dtype == np.float64 else lib().lslam_icp_large_starts
        check(entry(ptr(s32), n, ptr(t32), m, d, ptr(np.ascontiguousarray(init, dtype=np.float64)), S, float(epsilon),
                    float(stopping_thresh), int(max_iters), ICP_NORMALIZE_ERROR, ptr(poses), ptr(errors), ptr(iters), icp._DEVICE))
    else:
        r = icp_batch(_tiled([s32], S, S * n, d, 0), np.arange(S + 1, dtype=np.int64) * n, _tiled([t32], S, S * m, d, 1),
                      np.arange(S + 1, dtype=np.int64) * m, init, epsilon=epsilon, max_iters=max_iters,
                      stopping_thresh=stopping_thresh, normalize_error=True, device=icp._DEVICE)
        poses, errors, iters = r.T, r.error, r.iterations
    best = int(np.argmin(errors))                                # argmin = first index of the minimum = the reference's strict <
    out = (poses[best].copy(), np.float64(errors[best]))
    if return_all:
        out += (poses, errors, iters)
    return out


def _tiled(blocks, reps, total, d, slot):
    """every block ``reps`` times in a row, packed — in pinned staging slot ``slot`` when the library can provide it"""
    dtype = blocks[0].dtype
    try:
        from . import _lib
        out = _lib.staging_array(slot, total * d, dtype).reshape(total, d)
    except Exception:
        out = np.empty((total, d), dtype=dtype)
    at = 0
    for b in blocks:
        n = b.shape[0]
        out[at:at + reps * n].reshape(reps, n, d)[:] = b
        at += reps * n
    return out


def register_multi_start_many(source_pc, target_pcs, n_starts=24, epsilon=0.001, voxel_size=0.005,
                              downsample_above=20000, max_iters=2000, stopping_thresh=0.0001):
    """The outer loop of test_icp.py:66-92 as well: every measured cloud × every yaw start in ONE launch (one
    workgroup per run; with 4 views × 24 starts that is 96 compute units busy instead of 24).  Returns a list of
    ``(best_pose, best_error)``, one per target, each equal to what ``register_multi_start`` returns for it."""
    source_pc = np.asarray(source_pc)
    targets = [np.asarray(t) for t in target_pcs]
    d = 2 if source_pc.shape[1] == 2 and all(t.shape[1] == 2 for t in targets) else 3
    src_blocks, tgt_blocks, inits, n_src, n_tgt = [], [], [], [], []
    for t in targets:
        inits.append(yaw_starts(source_pc, t, n_starts))         # centroids of the FULL clouds (test_icp.py:72-74)
        if source_pc.shape[0] > downsample_above or t.shape[0] > downsample_above:
            s_ds, t_ds = icp.voxel_downsample(source_pc, voxel_size), icp.voxel_downsample(t, voxel_size)
        else:
            s_ds, t_ds = source_pc, t
        dtype = _cloud_dtype(source_pc, t)
        src_blocks.append(np.ascontiguousarray(icp._xyz(s_ds)[:, :d], dtype=dtype))
        tgt_blocks.append(np.ascontiguousarray(icp._xyz(t_ds)[:, :d], dtype=dtype))
        n_src.append(src_blocks[-1].shape[0]); n_tgt.append(tgt_blocks[-1].shape[0])
    if not targets:
        return []
    src_off = np.concatenate(([0], np.cumsum(np.repeat(n_src, n_starts)))).astype(np.int64)
    tgt_off = np.concatenate(([0], np.cumsum(np.repeat(n_tgt, n_starts)))).astype(np.int64)
    # the n_starts copies of every cloud are written straight into the library's pinned staging buffers: the upload is then
    # one DMA at PCIe speed (from a pageable vstack it ran at 6-10 GB/s, depending on the host: 2-4 ms for C1's 23 MB)
    src = _tiled(src_blocks, n_starts, int(src_off[-1]), d, 0)
    tgt = _tiled(tgt_blocks, n_starts, int(tgt_off[-1]), d, 1)
    r = icp_batch(src, src_off, tgt, tgt_off, np.concatenate(inits), epsilon=epsilon, max_iters=max_iters,
                  stopping_thresh=stopping_thresh, normalize_error=True, device=icp._DEVICE)
    out = []
    for v in range(len(targets)):
        e = r.error[v * n_starts:(v + 1) * n_starts]
        best = int(np.argmin(e))                                   # first index of the minimum = the reference's strict <
        out.append((r.T[v * n_starts + best].copy(), np.float64(e[best])))
    return out
