"""Batched 3-D problems (S x n x n) on the all-pairs path: 64-VGPR build against the 128-VGPR filter build."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from lidar_slam_amd import batch as B
rng = np.random.default_rng(1)
st = torch.cuda.current_stream().cuda_stream
for n, S in ((250, 4096), (500, 2048), (1000, 1024), (2000, 512), (4000, 256), (4000, 96)):
    t = rng.normal(size=(S, n, 3)); th = 0.05
    R = np.eye(3); R[0, 0] = R[1, 1] = np.cos(th); R[0, 1] = -np.sin(th); R[1, 0] = np.sin(th)
    s_ = (t - 0.03) @ R + rng.normal(0, 0.01, t.shape)
    ds = torch.from_numpy(s_.reshape(-1, 3).astype(np.float32)).cuda(); dt = torch.from_numpy(t.reshape(-1, 3).astype(np.float32)).cuda()
    off = torch.arange(S + 1, dtype=torch.int64, device="cuda") * n
    init = torch.eye(4, dtype=torch.float64, device="cuda").repeat(S, 1, 1).contiguous()
    oT = torch.empty((S, 4, 4), dtype=torch.float64, device="cuda"); oe = torch.empty(S, dtype=torch.float64, device="cuda"); oi = torch.empty(S, dtype=torch.int32, device="cuda")
    line = f"n={n} S={S}"
    for tag, env in (("all pairs", {}), ("pruned", {"PRUNE": "1"}), ("small", {"LSLAM_ICP_VARIANT": "2"}), ("big direct", {"LSLAM_ICP_VARIANT": "1", "LSLAM_ICP_FILTER": "0"})):
        prune = env.pop("PRUNE", None) is not None if "PRUNE" in env else False
        os.environ.update(env)
        def run():
            B.icp_batch_dev(ds.data_ptr(), off.data_ptr(), dt.data_ptr(), off.data_ptr(), 3, init.data_ptr(), S, n, n, oT.data_ptr(), oe.data_ptr(), oi.data_ptr(), None,
                            epsilon=0.0, max_iters=9, stopping_thresh=0.0, stream=st, prune=prune)
        for _ in range(2): run()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): run()
        torch.cuda.synchronize(); d = (time.perf_counter() - t0) / 5
        for k in env: os.environ.pop(k)
        line += f"  {tag} {d*1e3:7.3f} ms"
    print(line, flush=True)
