"""The parts of bench.py's contract that can be checked without a GPU: flags and defaults, the CPU baseline legs
(they are what `cpu_baseline` reports), and that a run without a GPU refuses instead of falling back."""
import os
import subprocess
import sys

import numpy as np

from conftest import ROOT

sys.path.insert(0, ROOT)


def test_flags_and_defaults(monkeypatch):
    import bench
    monkeypatch.setattr(sys, "argv", ["bench.py"])
    a = bench.parse()
    assert (a.gpus, a.workload, a.pairs, a.points, a.iters) == (1, "c4", 4096, 4096, 50)
    assert a.steps >= 1 and a.warmup >= 1
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "3", "--warmup", "1"])
    a = bench.parse()
    assert (a.gpus, a.steps, a.warmup) == (8, 3, 1)


def test_cpu_baseline_leg_runs_the_port_on_a_sample():
    import bench
    rng = np.random.default_rng(0)
    S, P, I = 3, 64, 5
    tgt = rng.uniform(-10, 10, (S * P, 2))
    src = tgt + rng.normal(0, 0.01, tgt.shape)
    init = np.tile(np.eye(4), (S, 1, 1))
    v, secs, its = bench.cpu_baseline_icp(src, tgt, init, P, S, I)
    assert its == S * I and v > 0 and secs > 0


def test_refuses_to_run_without_a_gpu():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "1"], capture_output=True, text=True,
                       env={**os.environ, "HIP_VISIBLE_DEVICES": "", "ROCR_VISIBLE_DEVICES": ""}, timeout=300)
    assert r.returncode != 0 and r.stdout.strip() == ""          # no JSON line, no CPU fallback
    assert "GPU" in r.stderr
