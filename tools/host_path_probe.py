"""C3 stage and OGM build_map from host lists: where the time goes (GPU box)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lidar_slam_amd import synth, batch as B, _lib
from lidar_slam_amd.modules.localization import scan_matching_relative_poses, chain_poses, relative_pose_guesses
from lidar_slam_amd.modules.ogm import OccupancyGridMap
seq = synth.sequence_c3(4956, seed=20)
z = seq.z_ts
def t(fn, reps=5):
    fn(); ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); ts.append((time.perf_counter() - t0) * 1e3)
    return f"{min(ts):.2f} ms (median {sorted(ts)[len(ts)//2]:.2f})"
print("stage (float64 list, staged pack):", t(lambda: chain_poses(scan_matching_relative_poses(seq.x_ts, z))))
z32 = [a.astype(np.float32) for a in z]
print("stage (float32 list, staged pack):", t(lambda: chain_poses(scan_matching_relative_poses(seq.x_ts, z32))))
print("  pack staged f64:", t(lambda: B.pack_clouds(z, staging=0)))
print("  pack numpy f64 :", t(lambda: B.pack_clouds(z)))
print("  pack numpy f32 :", t(lambda: B.pack_clouds(z, dtype=np.float32)))
init = relative_pose_guesses(seq.x_ts)
pts, off = B.pack_clouds(z, staging=0)
print("  icp_batch from staged f64:", t(lambda: B.icp_batch(pts, off[1:], pts, off[:-1], init, epsilon=0.001)))
p2, o2 = B.pack_clouds(z)
print("  icp_batch from pageable f64:", t(lambda: B.icp_batch(p2, o2[1:], p2, o2[:-1], init, epsilon=0.001)))
p3, o3 = B.pack_clouds(z, dtype=np.float32)
print("  icp_batch from pageable f32:", t(lambda: B.icp_batch(p3, o3[1:], p3, o3[:-1], init, epsilon=0.001)))
print("  guesses:", t(lambda: relative_pose_guesses(seq.x_ts)))
g = OccupancyGridMap(0.05, 30, 30, -30, -30)
print("ogm build_map (list, staged):", t(lambda: OccupancyGridMap(0.05, 30, 30, -30, -30).build_map(seq.true_poses, z), reps=3))
zs = B.PackedScans(*B.pack_clouds(z))
print("ogm build_map (PackedScans):", t(lambda: OccupancyGridMap(0.05, 30, 30, -30, -30).build_map(seq.true_poses, zs), reps=3))
print("ogm grid create only:", t(lambda: OccupancyGridMap(0.05, 30, 30, -30, -30), reps=3))
