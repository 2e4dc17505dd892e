"""Per-iteration latency of LiDAR-sized problems: S problems (one per workgroup slot or fewer) with a forced iteration
count; time / iterations = the length of one iteration's critical path inside a workgroup."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from lidar_slam_amd import synth, batch as B
from lidar_slam_amd.modules.localization import relative_pose_guesses
seq = synth.sequence_c3(1025, seed=20)
pts, off = B.pack_clouds([np.asarray(z) for z in seq.z_ts])
d_pts = torch.from_numpy(pts).cuda(); d_off = torch.from_numpy(off).cuda()
init = torch.from_numpy(relative_pose_guesses(seq.x_ts)).cuda()
maxn = int(np.diff(off).max())
st = torch.cuda.current_stream().cuda_stream
for S in (1, 64, 256, 512, 768, 1024):
    oT = torch.empty((S, 4, 4), dtype=torch.float64, device="cuda"); oe = torch.empty(S, dtype=torch.float64, device="cuda")
    oi = torch.empty(S, dtype=torch.int32, device="cuda")
    line = f"S={S:5d}"
    for prune in (True, False):
        for iters in (10, 40):
            def run():
                B.icp_batch_dev(d_pts.data_ptr(), d_off.data_ptr() + 8, d_pts.data_ptr(), d_off.data_ptr(), 2, init.data_ptr(), S, maxn, maxn,
                                oT.data_ptr(), oe.data_ptr(), oi.data_ptr(), None, epsilon=0.0, max_iters=iters - 1, stopping_thresh=0.0, stream=st, prune=prune)
            for _ in range(3): run()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(10): run()
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
            line += f"  {'pruned' if prune else 'all-pairs'} {iters} it: {dt*1e6:8.1f} us"
    print(line, flush=True)
