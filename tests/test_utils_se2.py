"""The product's SE(2)/SE(3) glue (lidar_slam_amd/modules/utils.py; reference modules/utils.py:140-240) against the
reference's committed outputs (outputs/poses_odom_20.npy -> relative_poses_odom_20.npy, slices in se2_outputs20.npz)
and against golden vectors emitted by the imported reference (tests/golden/utils_glue.npz, make_golden_utils.py)."""
import numpy as np
import pytest

from conftest import golden
from lidar_slam_amd.modules import utils as U


def test_get_relative_pose_reproduces_the_references_committed_outputs():
    g = golden("se2_outputs20.npz")
    p = g["poses_odom"]
    rel = np.array([U.get_relative_pose(p[i], p[i + 1]) for i in range(p.shape[0] - 1)])
    np.testing.assert_allclose(rel, g["relative_poses_odom"], atol=1e-13)


def test_pose_chain_of_the_product_reproduces_the_references_committed_poses():
    from lidar_slam_amd.modules.localization import chain_poses
    g = golden("se2_outputs20.npz")
    np.testing.assert_allclose(chain_poses(g["relative_poses_scan_matching"]), g["poses_scan_matching"], atol=1e-12)
    # the same chain through the glue functions, one step at a time (localization.py:127-130)
    T = np.eye(3)
    for k, R in enumerate(g["relative_poses_scan_matching"][:50]):
        T = T @ R
        np.testing.assert_allclose(U.pose_from_T(T), g["poses_scan_matching"][k + 1], atol=1e-12)


def test_glue_functions_equal_the_reference_bit_for_bit():
    g = golden("utils_glue.npz")
    for k in range(g["poses"].shape[0]):
        assert np.array_equal(U.T_from_pose(g["poses"][k]), g["T_from_pose"][k])
        assert np.array_equal(U.pose_from_T(g["T_from_pose"][k]), g["pose_from_T3"][k])
        assert np.array_equal(U.TSE3_from_TSE2(g["T_from_pose"][k]), g["TSE3_from_TSE2"][k])
        assert np.array_equal(U.TSE2_from_TSE3(g["T4"][k]), g["TSE2_from_TSE3"][k])
        assert np.array_equal(U.pose_from_T(g["T4"][k]), g["pose_from_T4"][k])
        if k + 1 < g["poses"].shape[0]:
            assert np.array_equal(U.get_relative_pose(g["poses"][k], g["poses"][k + 1]), g["relative"][k])
    assert np.array_equal(U.transform_points(g["pts3"], g["T4"][0]), g["pts3_T4"])
    assert np.array_equal(U.transform_points(g["pts2"], g["T_from_pose"][1]), g["pts2_T3"])


def test_transform_points_rejects_mismatched_shapes_like_the_reference():
    with pytest.raises(ValueError):
        U.transform_points(np.zeros((4, 3)), np.eye(3))
    with pytest.raises(ValueError):
        U.transform_points(np.zeros((4, 2)), np.eye(4))
