"""Workgroup size of the pruned planar batch kernel (LSLAM_ICP_PRUNE_THREADS) x points per lane (LSLAM_ICP_PRUNE_P) for batches of
different cloud sizes (C4-style clouds, natural convergence and 10 forced iterations) and for the C3 sequence."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from lidar_slam_amd import batch as B
dev = torch.device("cuda", 0)
st = torch.cuda.current_stream().cuda_stream

def timed(fn, warm=2, reps=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3

for n in (384, 768, 1536, 2048, 3072):
    S = 4096
    src, tgt, off, init, _ = bench.make_c4_on_device(torch, S, n, seed=7, device=dev)
    oT = torch.empty((S, 4, 4), dtype=torch.float64, device=dev); oe = torch.empty(S, dtype=torch.float64, device=dev)
    oi = torch.empty(S, dtype=torch.int32, device=dev)
    def run():
        B.icp_batch_dev(src.data_ptr(), off.data_ptr(), tgt.data_ptr(), off.data_ptr(), 2, init.data_ptr(), S, n, n,
                        oT.data_ptr(), oe.data_ptr(), oi.data_ptr(), None, epsilon=0.0, max_iters=9, stopping_thresh=0.0, stream=st, prune=True)
    out = []
    for P in ("1", "2"):
        for thr in ("0", "128", "256", "384", "512", "768"):
            os.environ["LSLAM_ICP_PRUNE_P"] = P
            if thr == "0": os.environ.pop("LSLAM_ICP_PRUNE_THREADS", None)
            else: os.environ["LSLAM_ICP_PRUNE_THREADS"] = thr
            out.append(f"P{P}/t{thr}: {timed(run):.3f}")
    print(f"n={n:5d} (10 forced iterations, ms)  " + "  ".join(out), flush=True)
