"""Pre-processing oracle vs the reference's get_lidar_data / DBSCAN / radial filter chain."""
import numpy as np

from conftest import golden, split
from oracle import preprocess_oracle as P


def test_chain_matches_reference():
    g = golden("lidar_pre.npz")
    z0 = P.lidar_xy(g["ranges"], float(g["range_min"]), float(g["range_max"]))
    for a, b in zip(z0, split(g["xy0"], g["off0"])):
        np.testing.assert_array_equal(a, b)
    z1 = P.dbscan_outliers_removal(split(g["xy0"], g["off0"]), eps=float(g["eps"]), min_samples=int(g["min_samples"]))
    for a, b in zip(z1, split(g["xy1"], g["off1"])):
        np.testing.assert_array_equal(a, b)
    z2 = P.statistical_outliers_removal(split(g["xy1"], g["off1"]), k_std=float(g["k_std"]))
    for a, b in zip(z2, split(g["xy2"], g["off2"])):
        np.testing.assert_array_equal(a, b)
