"""C5 (100k x 100k 3-D, 20 forced iterations) on the whole-chip path for different env settings / strategies."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from lidar_slam_amd import synth, batch as B
src, tgt, T_true, T0 = synth.cloud_c5()
s = torch.from_numpy(src).cuda(); t = torch.from_numpy(tgt).cuda()
st = torch.cuda.current_stream().cuda_stream
ref = None
for cfg in sys.argv[1:] or ["default"]:
    kvs = [] if cfg == "default" else cfg.split(",")
    prune = True if "PRUNE=1" in kvs else False
    for kv in kvs:
        k, v = kv.split("="); os.environ[k] = v
    res = {}
    def run():
        res["r"] = B.icp_large_dev(s.data_ptr(), s.shape[0], t.data_ptr(), t.shape[0], 3, T0, epsilon=0.0, max_iters=19, stopping_thresh=0.0, poll_every=20, stream=st, prune=prune)
    for _ in range(2): run()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): run()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    T = res["r"][0]
    if ref is None: ref = T
    print(f"{cfg:40s} {dt*1e3:8.3f} ms  {20*1e10/dt/1e12:6.2f} T pair-evals/s  max|dT| vs first {np.abs(T-ref).max():.2e}", flush=True)
    for kv in kvs: os.environ.pop(kv.split("=")[0])
