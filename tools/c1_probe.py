"""C1 (4 views x 24 yaw starts, 6000-pt model vs 4000-pt views): float32 vs float64-exact, pruned vs all pairs (GPU box)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lidar_slam_amd import synth
from lidar_slam_amd.icp_warm_up import register_multi_start_many
model, views, truths = synth.drill_c1()
for dt in (np.float32, np.float64):
    for prune in ("1", "0"):
        os.environ["LSLAM_ICP_PRUNE"] = prune
        m, v = model.astype(dt), [x.astype(dt) for x in views]
        register_multi_start_many(m, v)
        ts = []
        for _ in range(5):
            t0 = time.perf_counter(); r = register_multi_start_many(m, v); ts.append((time.perf_counter() - t0) * 1e3)
        print(dt.__name__, "prune", prune, f"{min(ts):.2f} ms", [f"{e:.3e}" for _, e in r], flush=True)
