"""C5 (100k x 100k 3-D, 20 forced iterations): float32 / float64-exact x all pairs / pruned (GPU box)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lidar_slam_amd import batch as B, synth
src, tgt, T_true, T0 = synth.cloud_c5()
s = torch.from_numpy(src).cuda(); t = torch.from_numpy(tgt).cuda(); s64 = s.double(); t64 = t.double()
st = torch.cuda.current_stream().cuda_stream
def run(f64, prune):
    return B.icp_large_dev(s.data_ptr(), s.shape[0], t.data_ptr(), t.shape[0], 3, T0, epsilon=0.0, max_iters=19, stopping_thresh=0.0, poll_every=20, stream=st, prune=prune,
                           d_src64=s64.data_ptr() if f64 else None, d_tgt64=t64.data_ptr() if f64 else None)
res = {}
for f64 in (False, True):
    for prune in (False, True):
        run(f64, prune); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(3): r = run(f64, prune)
        torch.cuda.synchronize(); res[(f64, prune)] = r
        print("C5 f64", f64, "prune", prune, f"{(time.perf_counter()-t0)/3*1e3:.2f} ms", r[2], flush=True)
print("f64 pruned vs f64 all pairs |dT|", np.abs(res[(True, True)][0] - res[(True, False)][0]).max(), "f32 pruned vs all pairs", np.abs(res[(False, True)][0] - res[(False, False)][0]).max())
