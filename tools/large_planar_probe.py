"""One large PLANAR pair on the whole-chip path (20 forced iterations): filter search against the direct sweep and pruning."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from lidar_slam_amd import batch as B
rng = np.random.default_rng(3)
st = torch.cuda.current_stream().cuda_stream
for n in (16000, 40000, 100000):
    t = rng.uniform(-20, 20, (n, 2)); th = 0.03
    R = np.array([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]])
    s_ = (t[rng.permutation(n)] - 0.05) @ R + rng.normal(0, 0.01, t.shape)
    ds = torch.from_numpy(s_.astype(np.float32)).cuda(); dt = torch.from_numpy(t.astype(np.float32)).cuda()
    line = f"n=m={n:7d}"
    ref = None
    for tag, env, prune in (("filter", None, False), ("direct", "0", False), ("pruned", None, True)):
        if env is not None: os.environ["LSLAM_ICP_FILTER"] = env
        res = {}
        def run():
            res["r"] = B.icp_large_dev(ds.data_ptr(), n, dt.data_ptr(), n, 2, np.eye(4), epsilon=0.0, max_iters=19, stopping_thresh=0.0, poll_every=20, stream=st, prune=prune)
        for _ in range(2): run()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(3): run()
        torch.cuda.synchronize(); dtm = (time.perf_counter() - t0) / 3
        os.environ.pop("LSLAM_ICP_FILTER", None)
        if ref is None: ref = res["r"][0]
        line += f"  {tag} {dtm*1e3:7.3f} ms (dT {np.abs(res['r'][0]-ref).max():.1e})"
    print(line, flush=True)
