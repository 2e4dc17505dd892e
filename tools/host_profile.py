"""cProfile of the host side of the two PCIe-inclusive stages (C3 scan matching, OGM build_map)."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from lidar_slam_amd import synth
from lidar_slam_amd.modules import localization as loc
from lidar_slam_amd.modules.ogm import OccupancyGridMap

seq = synth.sequence_c3(4956, seed=20)
z = [np.asarray(a) for a in seq.z_ts]
for _ in range(2):
    loc.poses_from_scan_matching(seq.x_ts, z, return_relative_poses=True)
t0 = time.perf_counter(); loc.poses_from_scan_matching(seq.x_ts, z, return_relative_poses=True); print("c3 stage ms", (time.perf_counter() - t0) * 1e3)
pr = cProfile.Profile(); pr.enable()
loc.poses_from_scan_matching(seq.x_ts, z, return_relative_poses=True)
pr.disable(); pstats.Stats(pr).sort_stats("cumulative").print_stats(18)

g = OccupancyGridMap(0.05, 30, 30, -30, -30); g.build_map(seq.true_poses[:8], z[:8])
g = OccupancyGridMap(0.05, 30, 30, -30, -30)
t0 = time.perf_counter(); g.build_map(seq.true_poses, z); print("ogm build_map ms", (time.perf_counter() - t0) * 1e3)
g = OccupancyGridMap(0.05, 30, 30, -30, -30)
pr = cProfile.Profile(); pr.enable()
g.build_map(seq.true_poses, z)
pr.disable(); pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
