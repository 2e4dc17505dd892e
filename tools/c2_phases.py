"""per-phase times of the co-resident C2 kernel (needs the -DLSLAM_COOP_PROFILE build: LSLAM_LIB=lidar_slam_amd/lib_prof/liblslam_hip.so)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lidar_slam_amd import batch as B, synth
dev = torch.device("cuda", 0)
st = torch.cuda.current_stream().cuda_stream
src, tgt, _ = synth.pair_c2(0)
s32 = torch.from_numpy(src.astype(np.float32)).to(dev); t32 = torch.from_numpy(tgt.astype(np.float32)).to(dev)
s64 = torch.from_numpy(src).to(dev); t64 = torch.from_numpy(tgt).to(dev)
for sub in ("1", "2"):
    os.environ["LSLAM_ICP_COOP_SUB"] = sub
    for f64 in (False, True):
        for rep in range(3):
            print(f"sub={sub} f64={f64}", file=sys.stderr, flush=True)
            B.icp_large_dev(s32.data_ptr(), 1081, t32.data_ptr(), 1081, 2, np.eye(4), epsilon=0.0, max_iters=49, stopping_thresh=0.0, stream=st, prune=False,
                            d_src64=s64.data_ptr() if f64 else None, d_tgt64=t64.data_ptr() if f64 else None)
