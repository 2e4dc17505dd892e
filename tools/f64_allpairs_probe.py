"""C4 all pairs in float64-exact mode for different launch settings (env LSLAM_ICP_*): python3 tools/f64_allpairs_probe.py [K=V,...] ..."""
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
def run():
    B.icp_batch_dev(src.data_ptr(), off.data_ptr(), tgt.data_ptr(), off.data_ptr(), 2, init.data_ptr(), S, P, P, oT.data_ptr(), oe.data_ptr(), oi.data_ptr(), None,
                    epsilon=0.0, max_iters=I - 1, stopping_thresh=0.0, stream=st, prune=False, d_src64=src64.data_ptr(), d_tgt64=tgt64.data_ptr())
ref = None
for cfg in sys.argv[1:] or ["default"]:
    for kv in ([] if cfg == "default" else cfg.split(",")):
        k, v = kv.split("="); os.environ[k] = v
    run(); torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); run(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    Th = oT.cpu().numpy()
    same = "first" if ref is None else ("identical" if np.array_equal(ref, Th) else f"max diff {np.abs(ref - Th).max():.3g}")
    ref = Th if ref is None else ref
    print(f"{cfg:50s} min {min(ts):7.2f} ms  frac of fp32 peak {S*I*P*P*5/(min(ts)*1e-3)/157.3e12:.3f}  T vs first: {same}", flush=True)
    for kv in ([] if cfg == "default" else cfg.split(",")):
        os.environ.pop(kv.split("=")[0])
