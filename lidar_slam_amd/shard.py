"""Sharding of a scan sequence over the GPUs of one node (one process per GPU).

The S frame-to-frame ICP problems of a sequence are independent work items — the initial guess of
pair i comes from the odometry poses only (reference: modules/localization.py:116-118) — so rank r
simply owns a block of pairs (contiguous, or every world-th pair when iteration counts drift along the
sequence) and there is NO data-path collective: the one exchange is
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


def owned_pairs(S: int, world: int, rank: int, assignment: str = "contiguous") -> np.ndarray:
    """Indices of the problems `rank` solves.  "contiguous": the block of `partition`; "interleaved":
    rank, rank + world, … — the even split when iteration counts vary slowly along the sequence
    (natural convergence: 2–15 iterations per pair)."""
    if assignment == "contiguous":
        lo, hi = partition(S, world, rank)
        return np.arange(lo, hi, dtype=np.int64)
    if assignment == "interleaved":
        return np.arange(rank, S, world, dtype=np.int64)
    raise ValueError("assignment must be 'contiguous' or 'interleaved'")


def all_gather_transforms(local_T, S: int, group=None, assignment: str = "contiguous"):
    """local_T: torch tensor (n_local, 4, 4) float64 on the backend's device → (S, 4, 4) on every
    rank, in problem order.  Blocks are padded to the largest block for the collective."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    owned = [owned_pairs(S, world, r, assignment) for r in range(world)]
    cap = max((o.shape[0] for o in owned), default=0)
    padded = torch.zeros((cap, 4, 4), dtype=torch.float64, device=local_T.device)
    padded[: local_T.shape[0]] = local_T
    out = torch.empty((world * cap, 4, 4), dtype=torch.float64, device=local_T.device)
    dist.all_gather_into_tensor(out, padded, group=group)
    if assignment == "contiguous":
        return torch.cat([out[r * cap: r * cap + owned[r].shape[0]] for r in range(world)], dim=0)
    full = torch.empty((S, 4, 4), dtype=torch.float64, device=local_T.device)
    for r in range(world):
        full[torch.from_numpy(owned[r]).to(full.device)] = out[r * cap: r * cap + owned[r].shape[0]]
    return full


def default_solver(device: int):
    """(scans_slice list, init_T (k,4,4)) -> (k,4,4): the HIP batch kernel on `device`."""
    from .batch import icp_batch, pack_clouds

    def solve(z_block, init_T):
        pts, off = pack_clouds([np.asarray(z) for z in z_block])
        return icp_batch(pts, off[1:], pts, off[:-1], init_T, epsilon=0.001, device=device).T
    return solve


def default_pair_solver(device: int):
    """(sources list, targets list, init_T (k,4,4)) -> (k,4,4) for pairs that are not consecutive in memory."""
    from .batch import icp_batch, pack_clouds

    def solve(sources, targets, init_T):
        src, src_off = pack_clouds([np.asarray(z) for z in sources])
        tgt, tgt_off = pack_clouds([np.asarray(z) for z in targets])
        return icp_batch(src, src_off, tgt, tgt_off, init_T, epsilon=0.001, device=device).T
    return solve


def poses_from_scan_matching_sharded(x_ts, z_ts, return_relative_poses=False, *, group=None,
                                     solver=None, device=None, assignment="contiguous"):
    """Distributed counterpart of modules/localization.py:95-134.  Every rank passes the full
    odometry poses and scans (host-side inputs, as in the reference) and gets the full result.
    `solver` defaults to the HIP kernel on this rank's GPU; the CPU tests inject the oracle.
    ``assignment="interleaved"`` gives rank r the pairs r, r + world, … (`solver` then takes
    ``(sources, targets, init_T)``)."""
    import torch
    import torch.distributed as dist

    from .modules.localization import chain_poses, relative_pose_guesses
    x_ts = np.asarray(x_ts, dtype=np.float64)
    S = x_ts.shape[0] - 1
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    own = owned_pairs(S, world, rank, assignment)
    on_gpu = dist.get_backend(group) == "nccl"
    if device is None:
        device = torch.cuda.current_device() if on_gpu else 0
    init = relative_pose_guesses(x_ts)[own]
    if own.shape[0] == 0:
        local = np.zeros((0, 4, 4))
    elif assignment == "contiguous":
        # pairs lo..hi-1 need scans lo..hi: one scan of overlap, no halo exchange (inputs are host-side)
        lo, hi = int(own[0]), int(own[-1]) + 1
        local = (solver or default_solver(device))(z_ts[lo:hi + 1], init)
    else:
        # pair i: source = scan i+1, target = scan i (localization.py:120-121)
        local = (solver or default_pair_solver(device))([z_ts[i + 1] for i in own], [z_ts[i] for i in own], init)
    t = torch.from_numpy(np.ascontiguousarray(local, dtype=np.float64).reshape(-1, 4, 4))
    if on_gpu:
        t = t.to(torch.device("cuda", device))
    T = all_gather_transforms(t, S, group, assignment).cpu().numpy()
    rel = np.tile(np.eye(3), (S, 1, 1))
    rel[:, :2, :2] = T[:, :2, :2]
    rel[:, :2, 2] = T[:, :2, 3]
    poses = chain_poses(rel)
    return (poses, rel) if return_relative_poses else poses
