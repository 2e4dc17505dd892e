"""float64-exact mode of the batched kernels against the reference's golden vectors (GPU box)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lidar_slam_amd.batch import icp_batch, pack_clouds
from oracle import c_oracle

G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
split = lambda cat, off: [cat[off[i]:off[i + 1]] for i in range(len(off) - 1)]

def report(name, r, gT, gcorr, giters, off):
    dT = np.abs(r.T - gT).max()
    mism = int((r.correspondences != gcorr).sum()) if gcorr is not None else -1
    print(f"{name}: max|dT| {dT:.2e}  corr mismatches {mism}  iters equal {np.array_equal(r.iterations, giters)}", flush=True)

for prune in (False, True):
    g = np.load(os.path.join(G, "icp_planar.npz"))
    srcs = split(g["src"], g["src_off"])
    flat, off = pack_clouds(srcs)
    S = len(srcs)
    tgt = np.tile(g["tgt"], (S, 1)); toff = np.arange(S + 1) * g["tgt"].shape[0]
    r = icp_batch(flat, off, tgt, toff, g["init"], epsilon=float(g["epsilon"]), return_correspondences=True, prune=prune)
    report(f"planar prune={prune}", r, g["T"], g["corr"], g["iterations"], off)
    r32 = icp_batch(flat.astype(np.float32), off, tgt.astype(np.float32), toff, g["init"], epsilon=float(g["epsilon"]), return_correspondences=True, prune=prune)
    report(f"planar f32 prune={prune}", r32, g["T"], g["corr"], g["iterations"], off)

    g = np.load(os.path.join(G, "icp_3d.npz")); m = np.load(os.path.join(G, "icp_multistart.npz"))
    from lidar_slam_amd import icp_warm_up as W
    init = W.yaw_starts(g["src"], g["tgt"])
    S = 24; n, mm = g["src"].shape[0], g["tgt"].shape[0]
    for dt in (np.float64, np.float32):
        src = np.tile(g["src"], (S, 1)).astype(dt); tg = np.tile(g["tgt"], (S, 1)).astype(dt)
        t0 = time.perf_counter()
        r = icp_batch(src, np.arange(S + 1) * n, tg, np.arange(S + 1) * mm, init, epsilon=0.001, normalize_error=True, return_correspondences=True, prune=prune)
        dt_s = time.perf_counter() - t0
        dR = np.abs(r.T[:, :3, :3] - m["T"][:, :3, :3]).reshape(S, -1).max(1)
        print(f"multistart {dt.__name__} prune={prune}: worst |dR| {dR.max():.2e} at start {dR.argmax()}, |dt| {np.abs(r.T[:, :3, 3] - m['T'][:, :3, 3]).max():.2e}, iters equal {np.array_equal(r.iterations, m['iterations'])}, {dt_s*1e3:.1f} ms", flush=True)

# random ragged problems vs the C oracle's float64 recipe
rng = np.random.default_rng(7)
for dim in (2, 3):
    ss, ts = [], []
    for n, m in [(8, 6), (70, 33), (1000, 999), (1081, 1081), (2500, 700), (513, 4096), (4096, 4096), (3000, 4000)]:
        t = rng.normal(size=(m, dim)) * 5.0
        th = rng.uniform(-0.2, 0.2); R = np.eye(dim); R[:2, :2] = [[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]]
        s = (t[rng.integers(0, m, n)] - rng.normal(0, 0.2, dim)) @ R + rng.normal(0, 0.05, (n, dim))
        if m > 40: t[m // 2] = t[3]
        ss.append(s); ts.append(t)
    fs, so = pack_clouds(ss); ft, to = pack_clouds(ts)
    I = np.tile(np.eye(4), (len(ss), 1, 1))
    for prune in (False, True):
        for var in ("auto", "big"):
            if var == "big": os.environ["LSLAM_ICP_VARIANT"] = "1"
            else: os.environ.pop("LSLAM_ICP_VARIANT", None)
            r = icp_batch(fs, so, ft, to, I, epsilon=1e-12, max_iters=10, return_correspondences=True, prune=prune)
            bad = []
            for k, (s, t) in enumerate(zip(ss, ts)):
                To, eo, co, ito = c_oracle.run_icp(s, t, np.eye(4), epsilon=1e-12, max_iters=10, recipe="f64")
                mm_ = int((r.correspondences[so[k]:so[k + 1]] != co).sum())
                if mm_ or r.iterations[k] != ito or np.abs(r.T[k] - To).max() > 1e-9:
                    bad.append((k, mm_, int(r.iterations[k]), ito, float(np.abs(r.T[k] - To).max())))
            print(f"random dim={dim} prune={prune} variant={var}: {'OK' if not bad else bad}", flush=True)
