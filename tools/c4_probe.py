"""Kernel-only timing of the C4 batch (S pairs x P x P planar, forced iterations) for different env settings (LSLAM_ICP_*).
   usage: python tools/c4_probe.py [cfg ...]   cfg = default | K=V[,K=V...] ; env C4_S, C4_P, C4_I, C4_NATURAL=1"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from lidar_slam_amd import batch as B
S = int(os.environ.get("C4_S", "4096")); P = int(os.environ.get("C4_P", "4096")); I = int(os.environ.get("C4_I", "50"))
NAT = os.environ.get("C4_NATURAL") == "1"
dev = torch.device("cuda", 0)
src, tgt, off, init, _ = bench.make_c4_on_device(torch, S, P, seed=1000, device=dev)
oT = torch.empty((S, 4, 4), dtype=torch.float64, device=dev); oe = torch.empty(S, dtype=torch.float64, device=dev)
oi = torch.empty(S, dtype=torch.int32, device=dev)
st = torch.cuda.current_stream().cuda_stream
def run(prune=False):
    B.icp_batch_dev(src.data_ptr(), off.data_ptr(), tgt.data_ptr(), off.data_ptr(), 2, init.data_ptr(), S, P, P,
                    oT.data_ptr(), oe.data_ptr(), oi.data_ptr(), None, epsilon=0.001 if NAT else 0.0,
                    max_iters=2000 if NAT else I - 1, stopping_thresh=1e-4 if NAT else 0.0, stream=st, prune=prune)
ref = None
for cfg in sys.argv[1:] or ["default"]:
    kvs = [] if cfg == "default" else cfg.split(",")
    prune = "PRUNE=1" in kvs
    for kv in kvs:
        k, v = kv.split("="); os.environ[k] = v
    for _ in range(2): run(prune)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    reps = 5
    for _ in range(reps): run(prune)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
    it = int(oi.sum().item())
    T = oT.cpu().numpy()
    if ref is None: ref = T
    print(f"{cfg:50s} {dt*1e3:8.3f} ms  iters {it}  {it*float(P)*P/dt/1e12:6.2f} T pair-evals/s  {it/dt/1e3:8.1f} k it/s  max|dT| vs first {np.abs(T-ref).max():.2e}", flush=True)
    for kv in kvs:
        os.environ.pop(kv.split("=")[0])
