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
