"""LiDAR pre-processing on the GPU vs the reference chain (golden) and the oracle."""
import numpy as np
import pytest

from conftest import golden, split

pytestmark = pytest.mark.gpu


def test_each_stage_matches_reference_exactly():
    from lidar_slam_amd.modules import localization as L
    g = golden("lidar_pre.npz")
    z0 = L.get_lidar_data(g["ranges"], float(g["range_min"]), float(g["range_max"]))
    for a, b in zip(z0, split(g["xy0"], g["off0"])):
        np.testing.assert_array_equal(a, b)                      # bit-exact points
    z1 = L.DBSCAN_outliers_removal(split(g["xy0"], g["off0"]), eps=float(g["eps"]), min_samples=int(g["min_samples"]))
    for a, b in zip(z1, split(g["xy1"], g["off1"])):
        np.testing.assert_array_equal(a, b)                      # same survivors, same order
    z2 = L.statistical_outliers_removal(split(g["xy1"], g["off1"]), k_std=float(g["k_std"]))
    for a, b in zip(z2, split(g["xy2"], g["off2"])):
        np.testing.assert_array_equal(a, b)


def test_fused_chain_matches_reference():
    from lidar_slam_amd.modules.localization import preprocess_lidar
    g = golden("lidar_pre.npz")
    scans, (mean, std, thr) = preprocess_lidar(g["ranges"], float(g["range_min"]), float(g["range_max"]),
                                               dbscan=(float(g["eps"]), int(g["min_samples"])), k_std=float(g["k_std"]),
                                               return_stats=True)
    for a, b in zip(scans, split(g["xy2"], g["off2"])):
        np.testing.assert_array_equal(a, b)
    d = np.sqrt(np.sum(g["xy1"] ** 2, axis=1))
    assert abs(mean - d.mean()) < 1e-12 and abs(std - d.std()) < 1e-12 and abs(thr - (d.mean() + 2 * d.std())) < 1e-12


def test_random_scans_against_oracle_including_empty_and_large():
    from lidar_slam_amd.modules.localization import preprocess_lidar
    from oracle import preprocess_oracle as P
    rng = np.random.default_rng(9)
    z = [rng.normal(size=(n, 2)) * s for n, s in ((0, 1), (1, 1), (9, 0.05), (300, 0.3), (5000, 2.0), (64, 0.02), (257, 0.5))]
    got = preprocess_lidar(z_ts=z, dbscan=(0.1, 10))
    exp = P.dbscan_outliers_removal(z, 0.1, 10)
    for a, b in zip(got, exp):
        np.testing.assert_array_equal(a, b)
    got = preprocess_lidar(z_ts=z, k_std=1.0)
    exp = P.statistical_outliers_removal(z, 1.0)
    for a, b in zip(got, exp):
        np.testing.assert_array_equal(a, b)
    # the DBSCAN passes skip groups of 16 consecutive points by their boxes: neighbours at a distance of exactly eps across a group
    # border (a line of points 0.1 apart), the same line shuffled, a wall-like curve in beam order and shuffled, clusters of
    # exactly min_samples points — every mask must still equal the full test's
    line = np.column_stack((np.arange(200) * 0.1, np.zeros(200)))
    th = np.linspace(-2.3, 2.3, 1081)
    wall = np.column_stack((4.0 * np.cos(th), 4.0 * np.sin(th))) + rng.normal(0, 0.003, (1081, 2))
    wall[::97] += 0.5                                                     # isolated returns: noise to be removed
    blobs = np.vstack([c + rng.normal(0, 0.01, (10, 2)) for c in rng.uniform(-3, 3, (40, 2))])
    z = [line, line[rng.permutation(200)], wall, wall[rng.permutation(1081)], blobs, blobs[rng.permutation(400)], line * [1.0000001, 1.0]]
    got = preprocess_lidar(z_ts=z, dbscan=(0.1, 10))
    exp = P.dbscan_outliers_removal(z, 0.1, 10)
    for a, b in zip(got, exp):
        np.testing.assert_array_equal(a, b)
    got = preprocess_lidar(z_ts=z, dbscan=(0.1, 2))
    exp = P.dbscan_outliers_removal(z, 0.1, 2)
    for a, b in zip(got, exp):
        np.testing.assert_array_equal(a, b)


def test_voxel_downsample_matches_reference_and_oracle():
    from lidar_slam_amd.modules import icp
    from oracle import icp_oracle
    g = golden("voxel.npz")
    out = icp.voxel_downsample(g["points"], float(g["voxel_size"]))
    assert out.shape == g["out"].shape
    np.testing.assert_allclose(out, g["out"], rtol=0, atol=1e-14)          # same voxels, same order, same means
    rng = np.random.default_rng(3)
    for n, d, v in ((1, 3, 0.1), (50000, 3, 0.005), (30000, 2, 0.05), (7, 2, 10.0)):
        pts = rng.normal(size=(n, d)) * 0.1
        got = icp.voxel_downsample(pts, v)
        exp = icp_oracle.voxel_downsample(pts, v)
        assert got.shape == exp.shape
        np.testing.assert_allclose(got, exp, rtol=0, atol=1e-14)
    with pytest.raises(ValueError):
        icp.voxel_downsample(rng.normal(size=(10, 3)) * 1e6, 1e-3)         # > 2^21 cells per axis


def test_resident_scan_set_pipeline_equals_host_pipeline():
    """ranges → filters → scan matching → occupancy grid with the scans staying in HBM gives exactly
    what the stage-by-stage host-array path gives"""
    from lidar_slam_amd.modules import localization as L
    from lidar_slam_amd.modules.ogm import OccupancyGridMap
    from lidar_slam_amd import synth
    seq = synth.sequence_c3(80, seed=3, drop=(0.0, 0.05), half=4.5, n_obstacles=6)
    host = L.preprocess_lidar(seq.ranges, 0.1, 30.0, dbscan=(0.1, 10), k_std=2)
    dev, stats = L.preprocess_lidar(seq.ranges, 0.1, 30.0, dbscan=(0.1, 10), k_std=2, resident=True, return_stats=True)
    assert len(dev) == len(host) == 80 and dev.n_points == sum(z.shape[0] for z in host)
    poses_h, rel_h = L.poses_from_scan_matching(seq.x_ts, host, return_relative_poses=True)
    poses_d, rel_d = L.poses_from_scan_matching(seq.x_ts, dev, return_relative_poses=True)
    np.testing.assert_array_equal(rel_d, rel_h)
    np.testing.assert_array_equal(poses_d, poses_h)
    gh, gd = OccupancyGridMap(0.05, 10, 10, -10, -10), OccupancyGridMap(0.05, 10, 10, -10, -10)
    gh.build_map(poses_h, host)
    gd.build_map(poses_d, dev)
    np.testing.assert_array_equal(gd.grid_map_log_odds, gh.grid_map_log_odds)
    np.testing.assert_array_equal(gd.grid_map, gh.grid_map)
    for a, b in zip(dev, host):                         # reading the set downloads the same points
        np.testing.assert_array_equal(a, b)
    plain = L.preprocess_lidar(z_ts=[np.asarray(z) for z in host], resident=True)      # no stage: just upload
    np.testing.assert_array_equal(plain[7], host[7])
