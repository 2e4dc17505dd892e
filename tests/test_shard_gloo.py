"""N > 1 path on CPU: world_size-2 gloo run of the sharded scan-matching driver, with the oracle
injected as the per-rank solver (the GPU solver is covered by the -m gpu tests)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

from conftest import ROOT, golden


def test_partition_covers_everything_once():
    from lidar_slam_amd.shard import block_sizes, partition
    for S in (0, 1, 7, 31, 4955, 4096):
        for world in (1, 2, 3, 8):
            blocks = [partition(S, world, r) for r in range(world)]
            assert blocks[0][0] == 0 and blocks[-1][1] == S
            assert all(blocks[i][1] == blocks[i + 1][0] for i in range(world - 1))
            assert max(block_sizes(S, world)) - min(block_sizes(S, world)) <= 1


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_interleaved_assignment_covers_everything_once():
    from lidar_slam_amd.shard import owned_pairs
    for S in (0, 1, 7, 4955):
        for world in (1, 2, 8):
            own = [owned_pairs(S, world, r, "interleaved") for r in range(world)]
            assert np.array_equal(np.sort(np.concatenate(own)), np.arange(S))
            assert max(o.shape[0] for o in own) - min(o.shape[0] for o in own) <= 1
    with pytest.raises(ValueError):
        owned_pairs(4, 2, 0, "random")


def _worker(rank, world, port, n_scans, out_dir, assignment="contiguous"):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from conftest import golden as g_, split as split_
    from lidar_slam_amd.shard import poses_from_scan_matching_sharded
    from oracle import icp_oracle
    g = g_("seq_scanmatch.npz")
    z = split_(g["z"], g["z_off"])[:n_scans]

    def oracle_solver(z_block, init_T):
        return np.array([icp_oracle.run_icp_full(z_block[i + 1], z_block[i], init_T[i], epsilon=0.001).T
                         for i in range(len(z_block) - 1)]).reshape(-1, 4, 4)

    def oracle_pair_solver(sources, targets, init_T):
        return np.array([icp_oracle.run_icp_full(s_, t_, T0, epsilon=0.001).T
                         for s_, t_, T0 in zip(sources, targets, init_T)]).reshape(-1, 4, 4)

    poses, rel = poses_from_scan_matching_sharded(g["x_ts"][:n_scans], z, return_relative_poses=True,
                                                  solver=oracle_solver if assignment == "contiguous" else oracle_pair_solver,
                                                  assignment=assignment)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), poses=poses, rel=rel)
    dist.destroy_process_group()


@pytest.mark.parametrize("n_scans,assignment", [(12, "contiguous"), (9, "contiguous"), (10, "interleaved")])   # even and ragged splits
def test_two_rank_gloo_matches_unsharded_reference(tmp_path, n_scans, assignment):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, n_scans, str(tmp_path), assignment), nprocs=2, join=True)
    g = golden("seq_scanmatch.npz")
    r0, r1 = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    assert np.array_equal(r0["poses"], r1["poses"]) and np.array_equal(r0["rel"], r1["rel"])
    np.testing.assert_allclose(r0["rel"], g["relative_poses"][: n_scans - 1], atol=1e-9)
    np.testing.assert_allclose(r0["poses"], g["poses"][:n_scans], atol=1e-8)
