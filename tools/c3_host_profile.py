"""Where the host-array C3 stage (poses_from_scan_matching on 4956 scans) spends its time: list packing, the library
call (host checks + PCIe + kernels), the chain — first call of the process and warm."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from lidar_slam_amd import synth, batch as B
from lidar_slam_amd.modules import localization as L
seq = synth.sequence_c3(4956, seed=20)
z = [np.asarray(s) for s in seq.z_ts]
def stage(tag):
    t0 = time.perf_counter(); pts, off = B.pack_clouds(z); t1 = time.perf_counter()
    g = L.relative_pose_guesses(seq.x_ts); t2 = time.perf_counter()
    r = B.icp_batch(pts, off[1:], pts, off[:-1], g, epsilon=0.001); t3 = time.perf_counter()
    rel = np.tile(np.eye(3), (len(z) - 1, 1, 1)); rel[:, :2, :2] = r.T[:, :2, :2]; rel[:, :2, 2] = r.T[:, :2, 3]
    poses = L.chain_poses(rel); t4 = time.perf_counter()
    print(f"{tag:8s} pack {1e3*(t1-t0):6.2f}  guesses {1e3*(t2-t1):6.2f}  icp_batch {1e3*(t3-t2):6.2f}  chain {1e3*(t4-t3):6.2f}  total {1e3*(t4-t0):6.2f} ms", flush=True)
L.scan_matching_relative_poses(seq.x_ts[:64], z[:64])
for tag in ("first", "warm", "warm", "warm"):
    stage(tag)
t0 = time.perf_counter(); L.poses_from_scan_matching(seq.x_ts, z); print(f"poses_from_scan_matching(list)   {1e3*(time.perf_counter()-t0):6.2f} ms")
zp = B.PackedScans(*(lambda p, o: (p.astype(np.float64), o))(*B.pack_clouds(z)))
t0 = time.perf_counter(); L.poses_from_scan_matching(seq.x_ts, zp); print(f"poses_from_scan_matching(packed) {1e3*(time.perf_counter()-t0):6.2f} ms")
