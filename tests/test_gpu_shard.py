"""The sharded scan-matching driver with the REAL per-rank solver: two processes share the one GPU of the box
(gloo carries the all-gather here; RCCL needs one GPU per rank, which the round-end 8-GPU run provides)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

from conftest import ROOT, golden, split

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, assignment, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from conftest import golden as g_, split as split_
    from lidar_slam_amd.shard import poses_from_scan_matching_sharded
    g = g_("seq_scanmatch.npz")
    z = split_(g["z"], g["z_off"])
    poses, rel = poses_from_scan_matching_sharded(g["x_ts"], z, return_relative_poses=True, device=0, assignment=assignment)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), poses=poses, rel=rel)
    dist.destroy_process_group()


@pytest.mark.parametrize("assignment", ["contiguous", "interleaved"])
def test_two_ranks_hip_solver_equals_single_process(tmp_path, assignment):
    from lidar_slam_amd.modules.localization import poses_from_scan_matching
    g = golden("seq_scanmatch.npz")
    z = split(g["z"], g["z_off"])
    poses1, rel1 = poses_from_scan_matching(g["x_ts"], z, return_relative_poses=True)
    mp.spawn(_worker, args=(2, _free_port(), assignment, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    assert np.array_equal(r0["rel"], r1["rel"]) and np.array_equal(r0["poses"], r1["poses"])
    np.testing.assert_array_equal(r0["rel"], rel1)             # a pair's result does not depend on who solved it
    np.testing.assert_array_equal(r0["poses"], poses1)


def _rccl_worker(rank, world, port, assignment, out_dir):
    """one rank on the RCCL backend: exercises the branch of shard.py that moves the local block to the GPU, gathers DEVICE
    tensors with all_gather_into_tensor and copies the result back — the code the round-end multi-GPU run executes per rank"""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    assert dist.get_backend() == "nccl"
    from conftest import golden as g_, split as split_
    from lidar_slam_amd import shard
    g = g_("seq_scanmatch.npz")
    z = split_(g["z"], g["z_off"])
    poses, rel = shard.poses_from_scan_matching_sharded(g["x_ts"], z, return_relative_poses=True, assignment=assignment)
    # the collective on its own, on device tensors of a ragged size
    S = 37
    own = shard.owned_pairs(S, world, rank, assignment)
    local = torch.arange(S * 16, dtype=torch.float64).reshape(S, 4, 4)[torch.from_numpy(own)].cuda()
    full = shard.all_gather_transforms(local, S, None, assignment)
    assert full.is_cuda and torch.equal(full.cpu(), torch.arange(S * 16, dtype=torch.float64).reshape(S, 4, 4))
    np.savez(os.path.join(out_dir, f"rccl{rank}.npz"), poses=poses, rel=rel)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("assignment", ["contiguous", "interleaved"])
def test_rccl_backend_one_rank_equals_single_process(tmp_path, assignment):
    """`backend="nccl"` (RCCL) with the one GPU of this box, in a fresh child process: bit for bit the single-process result"""
    from lidar_slam_amd.modules.localization import poses_from_scan_matching
    g = golden("seq_scanmatch.npz")
    z = split(g["z"], g["z_off"])
    poses1, rel1 = poses_from_scan_matching(g["x_ts"], z, return_relative_poses=True)
    mp.spawn(_rccl_worker, args=(1, _free_port(), assignment, str(tmp_path)), nprocs=1, join=True)
    r0 = np.load(tmp_path / "rccl0.npz")
    np.testing.assert_array_equal(r0["rel"], rel1)
    np.testing.assert_array_equal(r0["poses"], poses1)
