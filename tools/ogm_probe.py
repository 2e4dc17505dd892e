"""Times OccupancyGridMap.build_map on the synthetic dataset-20-like sequence (GPU) and the oracle on a sample."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from lidar_slam_amd import synth
from lidar_slam_amd.modules.ogm import OccupancyGridMap
from oracle.ogm_oracle import OccupancyGridOracle

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4956
seq = synth.sequence_c3(n, seed=20)
g = OccupancyGridMap(0.05, 30, 30, -30, -30)
g.build_map(seq.true_poses[:8], seq.z_ts[:8])      # warm-up
g = OccupancyGridMap(0.05, 30, 30, -30, -30)
t0 = time.perf_counter(); g.build_map(seq.true_poses, seq.z_ts); dt = time.perf_counter() - t0
rays = sum(z.shape[0] for z in seq.z_ts)
print(f"GPU build_map: {n} scans, {rays} rays: {dt*1e3:.1f} ms  -> {n/dt:.0f} scans/s, {rays/dt/1e6:.2f} M rays/s")
k = 24
o = OccupancyGridOracle(0.05, 30, 30, -30, -30)
t0 = time.perf_counter(); cells = 0
for i in range(k):
    c = o.scan_counts(seq.true_poses[i], seq.z_ts[i]); cells += int(np.abs(c).sum()); o.apply_counts(c)
dtc = time.perf_counter() - t0
print(f"oracle: {k} scans {dtc/k*1e3:.1f} ms/scan, {cells/k:.0f} cell updates/scan")
g2 = OccupancyGridMap(0.05, 30, 30, -30, -30); g2.build_map(seq.true_poses[:k], seq.z_ts[:k])
print("equal to oracle on sample:", np.array_equal(g2.grid_map_log_odds, o.grid_map_log_odds))
