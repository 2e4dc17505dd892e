"""Kernel-only timing of the C3 sequence batch for different launch geometries (env LSLAM_ICP_*)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from lidar_slam_amd import synth, batch as B
from lidar_slam_amd.modules.localization import relative_pose_guesses
REP = int(os.environ.get("C3_REP", "1"))
FORCE = int(os.environ.get("C3_FORCE_ITERS", "0"))
seq = synth.sequence_c3(4956, seed=20)
pts64, off = B.pack_clouds([np.asarray(z) for z in seq.z_ts] * REP, dtype=np.float64)
pts = pts64.astype(np.float32)
F64 = int(os.environ.get("C3_F64", "0"))
seq.x_ts = np.tile(seq.x_ts, (REP, 1))
S = len(seq.z_ts) * REP - 1
d_pts = torch.from_numpy(pts).cuda(); d_pts64 = torch.from_numpy(pts64).cuda(); d_off = torch.from_numpy(off).cuda()
init = torch.from_numpy(relative_pose_guesses(seq.x_ts)).cuda()
oT = torch.empty((S, 4, 4), dtype=torch.float64, device="cuda"); oe = torch.empty(S, dtype=torch.float64, device="cuda")
oi = torch.empty(S, dtype=torch.int32, device="cuda")
maxn = int(np.diff(off).max())
st = torch.cuda.current_stream().cuda_stream
def run():
    B.icp_batch_dev(d_pts.data_ptr(), d_off.data_ptr() + 8, d_pts.data_ptr(), d_off.data_ptr(), 2, init.data_ptr(), S, maxn, maxn,
                    oT.data_ptr(), oe.data_ptr(), oi.data_ptr(), None, epsilon=0.0 if FORCE else 0.001,
                    max_iters=(FORCE - 1) if FORCE else 2000, stopping_thresh=0.0 if FORCE else 1e-4, stream=st,
                    d_src64=d_pts64.data_ptr() + 16 * 0 if F64 else None, d_tgt64=d_pts64.data_ptr() if F64 else None)
n = np.diff(off)
for cfg in sys.argv[1:] or ["default"]:
    for kv in ([] if cfg == "default" else cfg.split(",")):
        k, v = kv.split("="); os.environ[k] = v
    for _ in range(3): run()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): run()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    it = oi.cpu().numpy()
    evals = float(np.sum(it.astype(np.float64) * n[1:] * n[:-1]))
    print(f"{cfg:60s} {dt*1e3:7.3f} ms  iters {it.sum()}  {evals/dt/1e12:.2f} T pair-evals/s")
    for kv in ([] if cfg == "default" else cfg.split(",")):
        os.environ.pop(kv.split("=")[0])
