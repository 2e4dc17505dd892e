"""SE(2) glue + scan-matching driver of the oracle vs the reference's committed outputs."""
import numpy as np

from conftest import golden, split
from oracle import se2_oracle as S


def test_relative_pose_reproduces_reference_outputs():
    g = golden("se2_outputs20.npz")
    p = g["poses_odom"]
    rel = np.array([S.get_relative_pose(p[i], p[i + 1]) for i in range(p.shape[0] - 1)])
    np.testing.assert_allclose(rel, g["relative_poses_odom"], atol=1e-13)


def test_chain_product_reproduces_reference_outputs():
    g = golden("se2_outputs20.npz")
    poses = S.chain_relative_poses(g["relative_poses_scan_matching"])
    np.testing.assert_allclose(poses, g["poses_scan_matching"], atol=1e-12)


def test_se2_se3_roundtrip():
    T3 = S.T_from_pose([1.0, -2.0, 0.3])
    assert np.array_equal(S.TSE2_from_TSE3(S.TSE3_from_TSE2(T3)), T3)
    np.testing.assert_allclose(S.pose_from_T(S.TSE3_from_TSE2(T3)), [1.0, -2.0, 0.3], atol=1e-15)


def test_lidar_xy_matches_reference_scan():
    g = golden("icp_planar.npz")
    r = np.load(__import__("os").path.join(__import__("conftest").GOLDEN, "test_ranges.npy"))
    np.testing.assert_allclose(S.lidar_xy_from_ranges(r, 0.1, 30), g["tgt"], atol=1e-15)


def test_scan_matching_sequence():
    g = golden("seq_scanmatch.npz")
    z = split(g["z"], g["z_off"])
    poses, rel, iters = S.poses_from_scan_matching(g["x_ts"], z, return_relative_poses=True,
                                                   return_iterations=True)
    assert iters.sum() == int(g["total_iterations"])
    np.testing.assert_allclose(rel, g["relative_poses"], atol=1e-9)
    np.testing.assert_allclose(poses, g["poses"], atol=1e-8)
