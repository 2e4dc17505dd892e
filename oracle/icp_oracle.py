"""NumPy restatement of the reference point-to-point ICP (test oracle).

TEST INFRASTRUCTURE — see ``oracle/__init__.py``.  Parity: PINNED by
``tests/golden/icp_*.npz`` (vectors emitted by the imported reference) —
``tests/test_oracle_icp.py`` replays every one of them.

Each function cites the reference lines it follows (paths relative to
``/root/reference``).  Two nearest-neighbour engines are offered:

* ``nn="kdtree"``  – ``scipy.spatial.KDTree`` exactly as ``modules/icp.py:40,161``
  (same complexity as the reference; this is what ``bench.py`` times as the
  CPU baseline);
* ``nn="brute"``   – all-pairs direct-difference argmin with lowest-index
  tie-break (the algorithm shape of the HIP kernels).

and two arithmetic recipes:

* ``recipe="f64"``    – everything float64, as the reference;
* ``recipe="device"`` – the HIP path's recipe: points rounded to float32,
  transformed source rounded to float32 for the search, distances evaluated in
  float32 as ``fma(dz,dz, fma(dy,dy, dx*dx))``, moments / solve / error in
  float64 on ``T @ float64(src32)``.  Lets the GPU tests demand *identical*
  correspondences and iteration counts instead of a tolerance.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np
from scipy.spatial import KDTree

__all__ = [
    "IcpResult", "as_xyz", "nn_bruteforce", "nn_bruteforce_f32", "kabsch_svd",
    "kabsch_planar_closed_form", "kabsch_from_moments", "sq_error",
    "icp_iteration", "run_icp", "run_icp_full", "voxel_downsample",
]


@dataclass
class IcpResult:
    T: np.ndarray                 # (4,4) float64
    error: float                  # error measured in the last executed iteration
    correspondences: np.ndarray   # (N,) int64, last executed iteration
    iterations: int               # number of icp iterations executed


def as_xyz(pc: np.ndarray) -> np.ndarray:
    """(N,2|3) -> (N,3) float64 with z=0 padding (modules/icp.py:153-156)."""
    pc = np.asarray(pc)
    if pc.ndim != 2 or pc.shape[1] not in (2, 3):
        raise ValueError("point cloud must have shape (N,2) or (N,3)")
    out = np.zeros((pc.shape[0], 3), dtype=np.float64)
    out[:, : pc.shape[1]] = pc
    return out


# --------------------------------------------------------------------------- NN
def nn_bruteforce(p: np.ndarray, q: np.ndarray, block: int = 512) -> np.ndarray:
    """Exact Euclidean 1-NN of every row of p among rows of q, float64,
    lowest index on ties.  Mathematically what ``KDTree.query(k=1)`` returns
    (modules/icp.py:29-41) up to the tie rule."""
    p = np.asarray(p, dtype=np.float64)
    q = np.asarray(q, dtype=np.float64)
    if q.shape[0] == 0:
        raise IndexError("empty target cloud")
    out = np.empty(p.shape[0], dtype=np.int64)
    for s in range(0, p.shape[0], block):
        d = p[s:s + block, None, :] - q[None, :, :]
        out[s:s + block] = np.argmin(np.einsum("ijk,ijk->ij", d, d), axis=1)
    return out


def _dist2_f32(p32: np.ndarray, q32: np.ndarray) -> np.ndarray:
    """float32 ``fma(dz,dz,fma(dy,dy,dx*dx))`` for all pairs; the fma is
    emulated through float64 (product of two float32 is exact in float64)."""
    d = p32[:, None, :] - q32[None, :, :]                 # float32, correctly rounded
    acc = (d[..., 0] * d[..., 0]).astype(np.float32)      # one rounding
    for k in range(1, p32.shape[1]):
        dk = d[..., k].astype(np.float64)
        acc = (dk * dk + acc.astype(np.float64)).astype(np.float32)
    return acc


def nn_bruteforce_f32(p32: np.ndarray, q32: np.ndarray, block: int = 256) -> np.ndarray:
    """Device-recipe NN: float32 direct-difference distances, lowest index on ties."""
    p32 = np.ascontiguousarray(p32, dtype=np.float32)
    q32 = np.ascontiguousarray(q32, dtype=np.float32)
    if q32.shape[0] == 0:
        raise IndexError("empty target cloud")
    out = np.empty(p32.shape[0], dtype=np.int64)
    for s in range(0, p32.shape[0], block):
        out[s:s + block] = np.argmin(_dist2_f32(p32[s:s + block], q32), axis=1)
    return out


# ----------------------------------------------------------------------- Kabsch
def kabsch_svd(p: np.ndarray, q: np.ndarray) -> np.ndarray:
    """Rigid transform taking p onto q (rows paired), modules/icp.py:43-74:
    centroids, S = Xc^T Yc, SVD, R = V diag(1,1,det(V U^T)) U^T, t = qbar - R pbar."""
    p = np.asarray(p, dtype=np.float64)[:, :3]
    q = np.asarray(q, dtype=np.float64)[:, :3]
    pbar = p.mean(axis=0)
    qbar = q.mean(axis=0)
    S = (p - pbar).T @ (q - qbar)
    U, _, Vt = np.linalg.svd(S)
    V = Vt.T
    D = np.diag([1.0, 1.0, np.linalg.det(V @ U.T)])
    R = V @ D @ U.T
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3] = qbar - R @ pbar
    return T


def kabsch_planar_closed_form(S2: np.ndarray) -> tuple[np.ndarray, float]:
    """Closed form of the xy block the reference's SVD yields for planar data
    (z == 0): the orthogonal polar factor of S2^T, a rotation if det S2 > 0 and
    a *reflection* otherwise, with R[2,2] = det(R2) (follows from
    modules/icp.py:62-67 when S has a zero third row and column).
    Returns (R2, R22)."""
    if S2[0, 0] * S2[1, 1] - S2[0, 1] * S2[1, 0] > 0.0:
        a = S2[0, 0] + S2[1, 1]
        b = S2[0, 1] - S2[1, 0]
        h = np.hypot(a, b)
        c, s = a / h, b / h
        return np.array([[c, -s], [s, c]]), 1.0
    a = S2[0, 0] - S2[1, 1]
    b = S2[0, 1] + S2[1, 0]
    h = np.hypot(a, b)
    if h == 0.0:                      # S2 == 0 (e.g. a single point): LAPACK gives I
        return np.eye(2), 1.0
    c, s = a / h, b / h
    return np.array([[c, s], [s, -c]]), -1.0


def kabsch_from_moments(n, sp, sq, spq, planar: bool) -> np.ndarray:
    """The device formulation: rigid transform from raw moments
    n, Σp, Σq, Σ p qᵀ (float64).  S = Σpqᵀ − Σp Σqᵀ / n  equals the centred
    cross-covariance of modules/icp.py:58-61."""
    sp = np.asarray(sp, dtype=np.float64)
    sq = np.asarray(sq, dtype=np.float64)
    S = np.asarray(spq, dtype=np.float64) - np.outer(sp, sq) / n
    pbar, qbar = sp / n, sq / n
    R = np.eye(3)
    if planar:
        R2, r22 = kabsch_planar_closed_form(S[:2, :2])
        R[:2, :2] = R2
        R[2, 2] = r22
    else:
        U, _, Vt = np.linalg.svd(S)
        V = Vt.T
        R = V @ np.diag([1.0, 1.0, np.linalg.det(V @ U.T)]) @ U.T
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3] = qbar - R @ pbar
    return T


# ------------------------------------------------------------------------ error
def sq_error(p: np.ndarray, q: np.ndarray, normalize: bool = False) -> float:
    """Σ‖p−q‖² (a sum, not a mean); with ``normalize`` divided by the squared
    diagonal of the joint bounding box times N (modules/icp.py:76-98)."""
    p = np.asarray(p, dtype=np.float64)
    q = np.asarray(q, dtype=np.float64)
    sse = np.sum((p - q) ** 2)
    if not normalize:
        return sse
    both = np.vstack((p, q))
    ext = both.max(axis=0) - both.min(axis=0)
    diag = np.sqrt(np.sum(ext ** 2))
    return sse / (diag ** 2 * p.shape[0])


# -------------------------------------------------------------------- iteration
def _transform(T: np.ndarray, xyz: np.ndarray) -> np.ndarray:
    """(T @ [xyz|1]^T)^T[:, :3]  (modules/icp.py:116)."""
    h = np.hstack((xyz, np.ones((xyz.shape[0], 1))))
    return (T @ h.T).T[:, :3]


def icp_iteration(src_xyz, tgt_xyz, T_prev, *, tree=None, normalize_error=False,
                  recipe="f64", planar=False):
    """One iteration (modules/icp.py:100-121): transform source by T_prev, match,
    solve the step, compose ``T_step @ T_prev``, and measure the error of the
    *pre-update* residual.  Returns (T_new, corr, error)."""
    p = _transform(T_prev, src_xyz)
    if recipe == "device":
        d = 2 if planar else 3
        corr = nn_bruteforce_f32(p[:, :d].astype(np.float32), tgt_xyz[:, :d].astype(np.float32))
    elif tree is not None:
        _, corr = tree.query(p)
    else:
        corr = nn_bruteforce(p, tgt_xyz)
    q = tgt_xyz[corr]
    if recipe == "device":
        n = p.shape[0]
        T_step = kabsch_from_moments(n, p.sum(axis=0), q.sum(axis=0), p.T @ q, planar)
    else:
        T_step = kabsch_svd(p, q)
    err = sq_error(p, q, normalize=normalize_error)
    return T_step @ T_prev, corr, err


def run_icp_full(pc1, pc2, init_transform=None, epsilon=0.01, max_iters=2000,
                 stopping_thresh=0.0001, normalize_error=False, *, nn="kdtree",
                 recipe="f64") -> IcpResult:
    """The convergence loop of modules/icp.py:123-189 with all outputs.

    Stop rules, in this order per iteration: ``error < epsilon``;
    ``iteration >= max_iters``; ``|last_err - error| < stopping_thresh`` (needs a
    previous error).  The returned transform is one step past the point where the
    returned error was measured (I1)."""
    planar = np.asarray(pc1).shape[1] == 2 and np.asarray(pc2).shape[1] == 2
    src = as_xyz(pc1)
    tgt = as_xyz(pc2)
    if recipe == "device":
        src = src.astype(np.float32).astype(np.float64)
        tgt = tgt.astype(np.float32).astype(np.float64)
    T = np.eye(4) if init_transform is None else np.array(init_transform, dtype=np.float64)
    if tgt.shape[0] == 0:
        raise IndexError("empty target cloud")          # reference: pc2[correspondences] on empty
    if not (np.all(np.isfinite(src)) and np.all(np.isfinite(tgt))):
        raise ValueError("'x' must be finite, check for nan or inf values")   # KDTree's message
    tree = KDTree(tgt) if (nn == "kdtree" and recipe == "f64") else None

    it = 0
    executed = 0
    last = None
    while True:
        T, corr, err = icp_iteration(src, tgt, T, tree=tree, normalize_error=normalize_error,
                                     recipe=recipe, planar=planar)
        executed += 1
        if err < epsilon:
            break
        if it >= max_iters:
            break
        if last is not None and abs(last - err) < stopping_thresh:
            break
        last = err
        it += 1
    return IcpResult(T=T, error=float(err), correspondences=np.asarray(corr, dtype=np.int64),
                     iterations=executed)


def run_icp(pc1, pc2, init_transform=None, epsilon=0.01, max_iters=2000,
            stopping_thresh=0.0001, return_error=False, normalize_error=False,
            return_correspondences=False, *, nn="kdtree", recipe="f64"):
    """Reference call surface and return convention (modules/icp.py:123-133,183-189)."""
    r = run_icp_full(pc1, pc2, init_transform, epsilon, max_iters, stopping_thresh,
                     normalize_error, nn=nn, recipe=recipe)
    if return_error and not return_correspondences:
        return r.T, r.error
    if return_correspondences and not return_error:
        return r.T, r.correspondences
    if return_error and return_correspondences:
        return r.T, r.error, r.correspondences
    return r.T


# -------------------------------------------------------------- voxel downsample
def voxel_downsample(point_cloud, voxel_size):
    """Mean of the points in each occupied voxel, voxels in lexicographic order of
    their integer index ``floor((p - min) / voxel_size)`` (modules/icp.py:4-27)."""
    pc = np.asarray(point_cloud)
    vox = np.floor((pc - pc.min(axis=0)) / voxel_size).astype(int)
    _, inverse, counts = np.unique(vox, axis=0, return_inverse=True, return_counts=True)
    inverse = inverse.reshape(-1)
    out = np.zeros((counts.shape[0], pc.shape[1]), dtype=np.float64)
    np.add.at(out, inverse, pc)
    return out / counts[:, None]
