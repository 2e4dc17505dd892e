"""bench.py's contract on the GPU box: one JSON line with the fields the driver reads, the N > 1 entry point, profile fields."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _run(*args, env=None, timeout=600):
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True,
                          env={**os.environ, **(env or {})}, timeout=timeout)


def test_small_run_prints_one_json_line_with_the_contract_fields():
    r = _run("--steps", "2", "--warmup", "1", "--pairs", "512", "--points", "1024", "--iters", "10", "--no-cpu-baseline", "--no-other-configs")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    b = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in b, key
    assert b["n_gpus"] == 1 and b["steps"] == 2 and b["warmup"] == 1 and b["dtype"] == "f32" and b["scaling"] == "weak"
    assert b["config"]["pairs_per_gpu"] == 512 and b["config"]["points"] == 1024 and "workload" in b["config"]
    rf = b["roofline"]
    assert rf["unit"] == "TFLOP/s" and 0.0 < rf["frac"] < 1.0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12
    # counters are profiled for the default shape only: another shape must not inherit them
    assert rf["traffic"] is None and rf["stale_profile"]
    assert b["value"] > 0 and abs(b["value"] - 512 * 10 * 2 / (b["ms_per_step"] * 2e-3)) < 1e-6 * b["value"]
    # the same batch through the library's default strategy and in float64-exact mode rides along
    assert b["default_search_strategy"]["max_abs_T_difference"] < 1e-9
    assert b["float64_exact_mode"]["ms_per_step_default_strategy"] > 0


def test_gpus_2_on_a_one_gpu_box_fails_loudly():
    import torch
    if torch.cuda.device_count() != 1:
        pytest.skip("needs a one-GPU box")
    r = _run("--gpus", "2", "--steps", "1", "--warmup", "1", "--pairs", "64", "--points", "256", "--no-cpu-baseline", "--no-other-configs")
    assert r.returncode != 0 and r.stdout.strip() == ""
    assert "rank 1" in r.stderr and "GPU" in r.stderr


def test_one_rank_rccl_rehearsal_reports_the_world_size_rccl_sees():
    r = _run("--steps", "1", "--warmup", "1", "--pairs", "256", "--points", "512", "--iters", "5", "--no-cpu-baseline", "--no-other-configs",
             env={"LSLAM_BENCH_FORCE_DIST": "1"})
    assert r.returncode == 0, r.stderr[-2000:]
    b = json.loads(r.stdout.strip().splitlines()[-1])
    assert b["n_gpus"] == 1 and "all-gather" in b["config"]["sharding"]
