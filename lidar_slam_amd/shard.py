"""Sharding of a scan sequence over the GPUs of one node (one process per GPU).

The S frame-to-frame ICP problems of a sequence are independent work items — the initial guess of
pair i comes from the odometry poses only (reference: modules/localization.py:116-118) — so rank r
simply owns a contiguous block of pairs and there is NO data-path collective: the one exchange is
an all-gather of the (S, 4, 4) float64 relative transforms (512 KiB at S = 4096; latency-bound over
xGMI), after which every rank can run the sequential float64 chain product
(localization.py:127-130).  ``torch.distributed`` is the transport: backend "nccl" (= RCCL) on the
GPU box, "gloo" in the CPU tests of this logic.
"""
from __future__ import annotations

import numpy as np


def partition(S: int, world: int, rank: int) -> tuple[int, int]:
    """Contiguous block [lo, hi) of the S problems owned by `rank`; sizes differ by at most one."""
    base, extra = divmod(S, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def block_sizes(S: int, world: int) -> list[int]:
    return [partition(S, world, r)[1] - partition(S, world, r)[0] for r in range(world)]


def all_gather_transforms(local_T, S: int, group=None):
    """local_T: torch tensor (n_local, 4, 4) float64 on the backend's device → (S, 4, 4) on every
    rank, in problem order.  Blocks are padded to the largest block for the collective."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    sizes = block_sizes(S, world)
    cap = max(sizes) if sizes else 0
    padded = torch.zeros((cap, 4, 4), dtype=torch.float64, device=local_T.device)
    padded[: local_T.shape[0]] = local_T
    out = torch.empty((world * cap, 4, 4), dtype=torch.float64, device=local_T.device)
    dist.all_gather_into_tensor(out, padded, group=group)
    return torch.cat([out[r * cap: r * cap + sizes[r]] for r in range(world)], dim=0)


def default_solver(device: int):
    """(scans_slice list, init_T (k,4,4)) -> (k,4,4): the HIP batch kernel on `device`."""
    from .batch import icp_batch, pack_clouds

    def solve(z_block, init_T):
        pts, off = pack_clouds([np.asarray(z) for z in z_block])
        return icp_batch(pts, off[1:], pts, off[:-1], init_T, epsilon=0.001, device=device).T
    return solve


def poses_from_scan_matching_sharded(x_ts, z_ts, return_relative_poses=False, *, group=None,
                                     solver=None, device=None):
    """Distributed counterpart of modules/localization.py:95-134.  Every rank passes the full
    odometry poses and scans (host-side inputs, as in the reference) and gets the full result.
    `solver` defaults to the HIP kernel on this rank's GPU; the CPU tests inject the oracle."""
    import torch
    import torch.distributed as dist

    from .modules.localization import chain_poses, relative_pose_guesses
    x_ts = np.asarray(x_ts, dtype=np.float64)
    S = x_ts.shape[0] - 1
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    lo, hi = partition(S, world, rank)
    on_gpu = dist.get_backend(group) == "nccl"
    if device is None:
        device = torch.cuda.current_device() if on_gpu else 0
    if solver is None:
        solver = default_solver(device)
    init = relative_pose_guesses(x_ts)[lo:hi]
    # pairs lo..hi-1 need scans lo..hi: one scan of overlap, no halo exchange (inputs are host-side)
    local = solver(z_ts[lo:hi + 1], init) if hi > lo else np.zeros((0, 4, 4))
    t = torch.from_numpy(np.ascontiguousarray(local, dtype=np.float64).reshape(-1, 4, 4))
    if on_gpu:
        t = t.to(torch.device("cuda", device))
    T = all_gather_transforms(t, S, group).cpu().numpy()
    rel = np.tile(np.eye(3), (S, 1, 1))
    rel[:, :2, :2] = T[:, :2, :2]
    rel[:, :2, 2] = T[:, :2, 3]
    poses = chain_poses(rel)
    return (poses, rel) if return_relative_poses else poses
