"""C4 default strategy (pruned batch kernel) for different launch settings (env LSLAM_ICP_*), float32 and float64-exact mode.
usage: python3 tools/prune_ab.py [K=V,K=V ...]"""
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
src64, tgt64 = src.double(), tgt.double()
oT = torch.empty((S, 4, 4), device=dev, dtype=torch.float64); oe = torch.empty(S, device=dev, dtype=torch.float64); oi = torch.empty(S, device=dev, dtype=torch.int32)
def run(f64):
    B.icp_batch_dev(src.data_ptr(), off.data_ptr(), tgt.data_ptr(), off.data_ptr(), 2, init.data_ptr(), S, P, P, oT.data_ptr(), oe.data_ptr(), oi.data_ptr(), None,
                    epsilon=0.0, max_iters=I - 1, stopping_thresh=0.0, stream=st, prune=True,
                    d_src64=src64.data_ptr() if f64 else None, d_tgt64=tgt64.data_ptr() if f64 else None)
ref = {}
for cfg in sys.argv[1:] or ["default"]:
    for kv in ([] if cfg == "default" else cfg.split(",")):
        k, v = kv.split("="); os.environ[k] = v
    for f64 in (False, True):
        run(f64); torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            t0 = time.perf_counter(); run(f64); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
        Th = oT.cpu().numpy()
        same = "first" if f64 not in ref else ("identical" if np.array_equal(ref[f64], Th) else f"DIFFERENT (max {np.abs(ref[f64] - Th).max():.3g})")
        ref.setdefault(f64, Th)
        print(f"{cfg:50s} {'float64-exact' if f64 else 'float32':14s} min {min(ts):7.2f} ms  median {sorted(ts)[2]:7.2f} ms  T vs first config: {same}", flush=True)
    for kv in ([] if cfg == "default" else cfg.split(",")):
        os.environ.pop(kv.split("=")[0])
