"""A/B of two builds of the library on the same box: the C4 headline launch, the default strategy, C3 and C2 (LSLAM_LIB picks the build).
usage: LSLAM_LIB=<path> python3 tools/ab_headline.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from lidar_slam_amd import batch as B
dev = torch.device("cuda", 0)
st = torch.cuda.current_stream().cuda_stream
S, P, I = 4096, 4096, 50
src, tgt, off, init, _ = bench.make_c4_on_device(torch, S, P, 1000, dev)
oT = torch.empty((S, 4, 4), device=dev, dtype=torch.float64); oe = torch.empty(S, device=dev, dtype=torch.float64); oi = torch.empty(S, device=dev, dtype=torch.int32)
def run(prune):
    B.icp_batch_dev(src.data_ptr(), off.data_ptr(), tgt.data_ptr(), off.data_ptr(), 2, init.data_ptr(), S, P, P, oT.data_ptr(), oe.data_ptr(), oi.data_ptr(), None,
                    epsilon=0.0, max_iters=I - 1, stopping_thresh=0.0, stream=st, prune=prune)
for name, prune, reps in (("all pairs", False, 4), ("default", None, 8)):
    run(prune); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); run(prune); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    print(os.environ.get("LSLAM_LIB", "product"), name, f"min {min(ts):.2f} ms  median {sorted(ts)[len(ts)//2]:.2f} ms", flush=True)
