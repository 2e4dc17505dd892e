"""Where does exact pruning start to pay?  Times both search strategies on (a) batches of planar problems of
n = m points (natural convergence, C4-style clouds scaled to constant density) and (b) one large pair on the whole
chip (20 forced iterations, planar and 3-D).  Feeds kAutoPruneBatchPoints / kAutoPruneLargePairEvals."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from lidar_slam_amd import batch as B
dev = torch.device("cuda", 0)
st = torch.cuda.current_stream().cuda_stream

def timed(fn, warm=2, reps=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3

print("batch, planar, natural convergence (epsilon 1e-3): ms all-pairs / ms pruned", flush=True)
for P in (64, 128, 256, 384, 512, 768, 1024, 2048, 4096):
    S = 4096
    src, tgt, off, init, _ = bench.make_c4_on_device(torch, S, P, seed=7, device=dev)
    oT = torch.empty((S, 4, 4), dtype=torch.float64, device=dev); oe = torch.empty(S, dtype=torch.float64, device=dev)
    oi = torch.empty(S, dtype=torch.int32, device=dev)
    res = {}
    for forced in (False, True):
        for prune in (False, True):
            def run():
                B.icp_batch_dev(src.data_ptr(), off.data_ptr(), tgt.data_ptr(), off.data_ptr(), 2, init.data_ptr(), S, P, P,
                                oT.data_ptr(), oe.data_ptr(), oi.data_ptr(), None, epsilon=0.0 if forced else 1e-3,
                                max_iters=9 if forced else 2000, stopping_thresh=0.0 if forced else 1e-4, stream=st, prune=prune)
            res[(forced, prune)] = timed(run)
    print(f"  n=m={P:5d}  natural {res[(False, False)]:8.3f} / {res[(False, True)]:8.3f}   10 forced iterations {res[(True, False)]:8.3f} / {res[(True, True)]:8.3f}  (its {int(oi.sum())})", flush=True)

print("whole chip, one pair, 20 forced iterations: ms all-pairs / ms pruned", flush=True)
rng = np.random.default_rng(3)
for dim in (2, 3):
    for n in (4000, 8000, 16000, 32000, 64000, 100000):
        t = rng.uniform(-1, 1, (n, dim)); t /= np.linalg.norm(t, axis=1, keepdims=True) ** 0.7      # a closed blob-like surface / curve
        t += rng.normal(0, 0.003, t.shape)
        th = 0.05; R = np.eye(dim); R[0, 0] = R[1, 1] = np.cos(th); R[0, 1] = -np.sin(th); R[1, 0] = np.sin(th)
        s_ = (t[rng.permutation(n)] - 0.02) @ R + rng.normal(0, 0.002, t.shape)
        ds = torch.from_numpy(s_.astype(np.float32)).to(dev); dt = torch.from_numpy(t.astype(np.float32)).to(dev)
        res = {}
        for prune in (False, True):
            def run():
                B.icp_large_dev(ds.data_ptr(), n, dt.data_ptr(), n, dim, np.eye(4), epsilon=0.0, max_iters=19, stopping_thresh=0.0,
                                poll_every=20, stream=st, prune=prune)
            res[prune] = timed(run, 2, 3)
        print(f"  dim {dim}  n=m={n:7d} ({n * n:.1e} pair evals/iteration)  {res[False]:8.3f} / {res[True]:8.3f}", flush=True)
