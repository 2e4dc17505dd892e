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


def test_launch_plan_decides_who_starts_the_ranks():
    """`--gpus N` without a launcher's environment must start N ranks itself (it used to run ONE rank silently);
    under torch.distributed.run the process is a rank; a contradiction between the two is an error."""
    import pytest
    import bench
    assert bench.launch_plan(1, {}) == "single"
    assert bench.launch_plan(8, {}) == "spawn"
    assert bench.launch_plan(2, {"PATH": "/bin"}) == "spawn"
    assert bench.launch_plan(8, {"WORLD_SIZE": "8", "RANK": "3"}) == "rank"
    assert bench.launch_plan(1, {"WORLD_SIZE": "4", "RANK": "0"}) == "rank"          # flag left at its default under a launcher
    assert bench.launch_plan(1, {"WORLD_SIZE": "1", "RANK": "0"}) == "single"
    with pytest.raises(SystemExit):
        bench.launch_plan(4, {"WORLD_SIZE": "8", "RANK": "0"})
    with pytest.raises(SystemExit):
        bench.launch_plan(0, {})


def test_gpus_2_without_two_gpus_fails_loudly_instead_of_measuring_one():
    """no GPU here: every spawned rank must refuse, the parent must relay the failure and print no JSON line"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1"],
                       capture_output=True, text=True, env={**os.environ, "HIP_VISIBLE_DEVICES": "", "ROCR_VISIBLE_DEVICES": ""}, timeout=600)
    assert r.returncode != 0 and r.stdout.strip() == ""
    assert "rank" in r.stderr and "GPU" in r.stderr


def test_spawned_ranks_get_a_launcher_environment_and_rank0_line_is_relayed(tmp_path):
    """spawn_ranks() with a stand-in rank program: N fresh processes, RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* set,
    rank 0's one line relayed on stdout, a failing rank makes the whole call fail"""
    prog = tmp_path / "fake_rank.py"
    prog.write_text("import os, sys, json\n"
                    "r = int(os.environ['RANK']); assert os.environ['LOCAL_RANK'] == str(r)\n"
                    "assert os.environ['MASTER_ADDR'] == '127.0.0.1' and int(os.environ['MASTER_PORT']) > 0\n"
                    "open(os.path.join(os.path.dirname(__file__), f'seen{r}'), 'w').write(os.environ['WORLD_SIZE'])\n"
                    "if r == 0: print(json.dumps({'n_gpus': int(os.environ['WORLD_SIZE'])}))\n"
                    "sys.exit(int(os.environ.get('FAIL_RANK', '-1')) == r)\n")
    code = subprocess.run([sys.executable, "-c",
                           f"import sys; sys.path.insert(0, {ROOT!r}); import bench; bench.__file__ = {str(prog)!r}; "
                           "sys.exit(bench.spawn_ranks(3, []))"], capture_output=True, text=True, timeout=120)
    assert code.returncode == 0 and code.stdout.strip() == '{"n_gpus": 3}'
    assert sorted(f.name for f in tmp_path.glob("seen*")) == ["seen0", "seen1", "seen2"]
    bad = subprocess.run([sys.executable, "-c",
                          f"import sys; sys.path.insert(0, {ROOT!r}); import bench; bench.__file__ = {str(prog)!r}; "
                          "sys.exit(bench.spawn_ranks(3, []))"], capture_output=True, text=True, timeout=120, env={**os.environ, "FAIL_RANK": "2"})
    assert bad.returncode != 0 and bad.stdout.strip() == "" and "rank 2" in bad.stderr


def test_profile_derived_fields_are_reported_only_for_the_sources_they_were_taken_on(tmp_path):
    """roofline.traffic / instruction counts come from a committed PMC profile: they must vanish (with a reason) as soon as
    the kernel sources differ from the profiled ones, or another kernel is launched"""
    import json
    import bench
    d = tmp_path / "profiles" / "r09"
    d.mkdir(parents=True)
    doc = {"csrc_sha256": "abc", "c4": {"kernel": "K<2,8,false>", "hbm_bytes_per_launch": 123.0, "lane_ops_per_pair_eval": 3.0}}
    (d / "pmc_traffic.json").write_text(json.dumps(doc))
    ok = bench.profile_figures("c4", "K<2,8,false>", root=str(tmp_path), current_hash="abc")
    assert ok["hbm_bytes_per_launch"] == 123.0 and ok["lane_ops_per_pair_eval"] == 3.0 and "stale_profile" not in ok
    stale = bench.profile_figures("c4", "K<2,8,false>", root=str(tmp_path), current_hash="other")
    assert "stale_profile" in stale and "hbm_bytes_per_launch" not in stale
    wrong_kernel = bench.profile_figures("c4", "K<2,4,false>", root=str(tmp_path), current_hash="abc")
    assert "stale_profile" in wrong_kernel and "hbm_bytes_per_launch" not in wrong_kernel
    assert "stale_profile" in bench.profile_figures("c4", "K", root=str(tmp_path / "nothing"), current_hash="abc")
    # a newer round without the hash (or without the entry) never lets an older one through
    d2 = tmp_path / "profiles" / "r10"
    d2.mkdir()
    (d2 / "pmc_traffic.json").write_text(json.dumps({"c4": doc["c4"]}))
    assert "stale_profile" in bench.profile_figures("c4", "K<2,8,false>", root=str(tmp_path), current_hash="abc")


def test_csrc_hash_follows_the_sources(tmp_path):
    from lidar_slam_amd._lib import csrc_hash
    h = csrc_hash()
    assert len(h) == 64 and h == csrc_hash()
