"""C2 (one 1081 x 1081 planar pair, 50 forced iterations): the co-resident one-launch path against the launch-per-iteration path,
float32 and float64-exact, checked against the C oracle (GPU box)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lidar_slam_amd import batch as B, synth
from oracle import c_oracle

dev = torch.device("cuda", 0)
st = torch.cuda.current_stream().cuda_stream
src, tgt, T_true = synth.pair_c2(0)
s32 = torch.from_numpy(src.astype(np.float32)).to(dev); t32 = torch.from_numpy(tgt.astype(np.float32)).to(dev)
s64 = torch.from_numpy(src).to(dev); t64 = torch.from_numpy(tgt).to(dev)
corr = torch.empty(src.shape[0], dtype=torch.int32, device=dev)
I = 50

def run(f64, iters=I, eps=0.0, thresh=0.0):
    return B.icp_large_dev(s32.data_ptr(), 1081, t32.data_ptr(), 1081, 2, np.eye(4), epsilon=eps, max_iters=iters - 1, stopping_thresh=thresh,
                           poll_every=iters, stream=st, prune=False, d_out_corr=corr.data_ptr(),
                           d_src64=s64.data_ptr() if f64 else None, d_tgt64=t64.data_ptr() if f64 else None)

def timed(f64, reps=30):
    for _ in range(3): run(f64)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): run(f64)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3

ref32 = c_oracle.run_icp(src, tgt, np.eye(4), epsilon=0.0, stopping_thresh=0.0, max_iters=I - 1, recipe="device")
ref64 = c_oracle.run_icp(src, tgt, np.eye(4), epsilon=0.0, stopping_thresh=0.0, max_iters=I - 1, recipe="f64")
for coop in ("1", "0"):
    os.environ["LSLAM_ICP_COOP"] = coop
    for sub in (("1", "2", "4") if coop == "1" else ("-",)):
        if sub != "-": os.environ["LSLAM_ICP_COOP_SUB"] = sub
        for f64 in ((False, True) if coop == "1" else (False,)):
            T, err, it = run(f64)
            torch.cuda.synchronize()
            ref = ref64 if f64 else ref32
            ok = it == ref[3] and np.array_equal(corr.cpu().numpy(), ref[2]) and np.abs(T - ref[0]).max() < 1e-9
            ms = timed(f64)
            print(f"coop={coop} sub={sub} f64={f64}: {ms:.3f} ms for {I} iterations = {ms * 1e3 / I:.2f} us/iteration, matches oracle: {ok} (|dT| {np.abs(T - ref[0]).max():.1e}, iters {it})", flush=True)
# natural convergence, single call latency from device pointers
os.environ["LSLAM_ICP_COOP"] = "1"; os.environ.pop("LSLAM_ICP_COOP_SUB", None)
for f64 in (False, True):
    T, err, it = run(f64, iters=2001, eps=1e-3, thresh=1e-4)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): run(f64, iters=2001, eps=1e-3, thresh=1e-4)
    torch.cuda.synchronize()
    print(f"natural convergence f64={f64}: {it} iterations, {(time.perf_counter() - t0) / 50 * 1e6:.1f} us per call", flush=True)
