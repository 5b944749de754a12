"""Host-side logic of bench.py that needs no GPU: the step schedule, the usable-thread count of the CPU baseline, and
how the offline counter profiles (profiles/*.json) are attached to a live line."""
import json
import math
import os
import sys

import pytest

from conftest import REPO

sys.path.insert(0, REPO)
import bench  # noqa: E402


def test_schedule_is_exactly_k_steps():
    for k in (1, 5, 20, 63, 64, 65, 128, 200, 4096):
        g = bench.groups_of(k, 64)
        assert sum(g) == k and all(0 < n <= 64 for n in g) and g[:-1] == [64] * (len(g) - 1)
    assert bench.groups_of(0, 64) == []


def test_host_threads_is_positive_and_within_the_affinity_mask():
    n = bench.host_threads()
    assert 1 <= n <= len(os.sched_getaffinity(0))


def test_every_workload_is_well_formed():
    for name, w in bench.WORKLOADS.items():
        assert {"E", "D", "phys", "ctrl", "act", "task"} <= set(w), name
        assert 240 % w["ctrl"] == 0 and 0 <= w["phys"] <= 7
    # BASELINE.json configs 4 and 5 exist as named per-GPU workloads
    assert bench.WORKLOADS["hover65536x8_allgather"]["allgather"] and bench.WORKLOADS["hover65536x8_allgather"]["E"] == 65536
    assert bench.WORKLOADS["multihover2x16384x8"]["D"] == 2 and bench.WORKLOADS["multihover2x16384x8"]["phys"] & 4


class _Core:
    N = 65536


def test_counter_profiles_scale_with_the_launch(tmp_path, monkeypatch):
    prof = tmp_path / "profiles"
    prof.mkdir()
    (prof / "hbm_traffic.json").write_text(json.dumps({
        "w:rollout64": {"traffic_bytes": 306.0e6, "algorithmic_bytes": 300.0e6, "env_steps_per_launch": 64, "rocprof_kernel_avg_ns": 56000.0}}))
    (prof / "kernel_counters.json").write_text(json.dumps({
        "w:rollout64": {"slots_per_wave_env_step": 300.0, "valu_per_wave_env_step": 250.0}}))
    monkeypatch.setattr(bench, "REPO", str(tmp_path))
    m = {"us_per_step": 0.9}
    # the profiled launch shape: counters as measured
    roof = {"env_steps_per_launch": 64.0, "bytes_per_launch": 300.0e6}
    issue = bench.attach_counters(roof, "w:rollout64", m, _Core, clock_ghz=2.0)
    assert roof["traffic"] == 306.0e6 and roof["rocprof_kernel_avg_us"] == 56.0
    assert issue["bound"] == "valu_issue" and issue["waves_per_simd"] == 1
    assert issue["floor_us"] == pytest.approx(300 * 4 / 2400.0) and issue["frac"] == pytest.approx(0.5 / 0.9)
    assert issue["frac_at_measured_clock"] == pytest.approx(0.6 / 0.9)
    assert roof["floor_us"] == pytest.approx(300.0e6 / 64 / 8000e3) and roof["binding"] == "hbm"
    # a 20-step launch (what `--steps 20` times): the 64-step counters are scaled by the algorithmic bytes, not pasted
    roof = {"env_steps_per_launch": 20.0, "bytes_per_launch": 99.0e6}
    bench.attach_counters(roof, "w:rollout64", m, _Core, clock_ghz=None)
    assert roof["traffic"] == pytest.approx(306.0 / 300.0 * 99.0e6) and roof["rocprof_kernel_avg_us"] is None
    assert "scaled" in roof["traffic_note"]
    # no profile for this key: nothing is invented
    roof = {"env_steps_per_launch": 1.0, "bytes_per_launch": 1.0e6}
    assert bench.attach_counters(roof, "other:graph", m, _Core, None) is None and roof.get("traffic") is None


# ---- `python bench.py --gpus N` starts its N ranks itself -------------------------------------------------
def test_self_launch_builds_a_torchrun_job_with_the_same_arguments(monkeypatch):
    import subprocess
    import torch
    seen = {}

    def fake_run(cmd, env=None, **kw):
        seen["cmd"], seen["env"] = cmd, env
        return subprocess.CompletedProcess(cmd, 0)

    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.setattr(torch.cuda, "is_available", lambda: True)
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 8)
    argv = ["--gpus", "4", "--steps", "20", "--warmup", "5"]
    with pytest.raises(SystemExit) as e:
        bench.self_launch(bench.parse_args(argv), argv)
    assert e.value.code == 0
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and 1024 < int(cmd[cmd.index("--master-port") + 1]) < 65536
    i = cmd.index(os.path.join(REPO, "bench.py"))
    assert cmd[i + 1:] == argv                                   # the ranks see exactly what the caller typed
    assert seen["env"]["MASTER_ADDR"] == "127.0.0.1" and seen["env"]["GPD_BENCH_SELF_LAUNCHED"] == "1"
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_self_launch_refuses_to_call_a_smaller_run_n_gpus(monkeypatch, capsys):
    import torch
    monkeypatch.setattr(torch.cuda, "is_available", lambda: True)
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 1)
    monkeypatch.delenv("GPD_BENCH_SINGLE_DEVICE", raising=False)
    argv = ["--gpus", "8"]
    with pytest.raises(SystemExit) as e:
        bench.self_launch(bench.parse_args(argv), argv)
    assert e.value.code == 2 and "refusing" in capsys.readouterr().err


def test_gpus_n_without_torchrun_never_runs_one_rank_silently():
    """The driver's command form on a box without N devices: a loud non-zero exit, no JSON line claiming anything."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "GPD_BENCH_SINGLE_DEVICE")}
    res = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1"],
                         cwd=REPO, env=env, capture_output=True, text=True, timeout=300)
    import torch
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        pytest.skip("this box really has two devices")
    assert res.returncode == 2 and not [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert "--gpus 2" in res.stderr and "refusing" in res.stderr


def test_suite_names_are_workloads_and_cover_baseline_configs_4_and_5():
    assert set(bench.SUITE) <= set(bench.WORKLOADS)
    assert set(bench.SUITE) == {"hover65536x8_allgather", "multihover2x16384x8"}
    a = bench.parse_args([])
    assert a.gpus == 1 and not a.scale_suite and not a.no_suite and a.suite_timeout > 0 and not a.dry_run_topology and a.init_timeout > 0


def test_the_driver_run_file_is_the_headline_and_its_legs_only():
    """bench.py is what the driver runs -- the headline, one_launch_per_step, hbm_saturating, dropin_single_env, parity,
    cpu_baseline, the suite for N > 1 -- in at most 900 lines; the experiments of rounds 3-5 are gone from it, the one-world / policy /
    history-row workloads live in bench_extra.py (which registers them into the same table and runs bench.main)."""
    text = open(os.path.join(REPO, "bench.py")).read()
    assert text.count("\n") <= 900
    for flag in ("--split", "--cu-mask", "--stagger", "--rollout-graph"):
        assert flag not in text
    assert not [n for n in bench.WORKLOADS if n.startswith("swarm") or "policy" in n] or "bench_extra" in sys.modules
    import bench_extra  # noqa: F401
    for name in ("swarm65536_ext_240hz", "swarm1m_ext_240hz", "hover65536_30hz_policy", "hover65536_240hz_fullobs", "hover65536_30hz_history"):
        w = bench.WORKLOADS[name]
        assert callable(w["builder"]) and {"E", "D", "phys", "ctrl", "act", "task"} <= set(w)
    assert bench.WORKLOADS["swarm1m_ext_240hz"]["scaling"] == "strong"
    # the checker code sits with the oracle, and bench.py reaches it only inside its parity / cpu_baseline legs
    import ast
    tree = ast.parse(text)
    top = [n for n in tree.body if isinstance(n, (ast.Import, ast.ImportFrom))]
    assert not [n for n in top if "oracle" in (getattr(n, "module", "") or "") or any("oracle" in a.name for a in n.names)]


def test_watchdog_prints_the_line_it_holds_and_leaves():
    """An extra leg that never returns (a rank waiting in a collective another rank left): rank 0 prints the headline with a
    note, every rank exits 0; a leg that finishes in time prints nothing."""
    import subprocess
    code = ("import sys, time, types; sys.path.insert(0, %r); import bench\n"
            "out = {'metric': 'm', 'value': 1.0}\n"
            "job = types.SimpleNamespace(rank=int(sys.argv[1]))\n"
            "quick = bench.Watchdog(0.2, job, out, lambda o: o.__setitem__('quick', 'fired')); quick.done(); time.sleep(0.4)\n"
            "bench.Watchdog(0.3, job, out, lambda o: o.__setitem__('extra', {'error': 'late'}))\n"
            "time.sleep(60)\n" % REPO)
    for rank, lines in ((0, 1), (1, 0)):
        res = subprocess.run([sys.executable, "-c", code, str(rank)], capture_output=True, text=True, timeout=120)
        got = [l for l in res.stdout.splitlines() if l.startswith("{")]
        assert res.returncode == 0 and len(got) == lines, (res.stdout, res.stderr)
        if lines:
            assert json.loads(got[0]) == {"metric": "m", "value": 1.0, "extra": {"error": "late"}}


def test_committed_counter_profiles_hold_the_keys_the_bench_line_reads():
    """`roofline.traffic` and `roofline_valu_issue` of the live line come from these files (a partial profile run once left them
    empty and the line silently lost both blocks)."""
    t = json.load(open(os.path.join(REPO, "profiles", "hbm_traffic.json")))
    c = json.load(open(os.path.join(REPO, "profiles", "kernel_counters.json")))
    s = json.load(open(os.path.join(REPO, "profiles", "swarm_counters.json")))
    for key in ("hover65536_240hz:rollout64", "hover65536_240hz:graph", "hover4m_240hz:rollout64"):
        assert t[key]["traffic_bytes"] > 0 and t[key]["algorithmic_bytes"] > 0, key
        # traffic within a few per cent of the algorithmic bytes: no wasted re-reads
        assert 0.9 < t[key]["traffic_bytes"] / t[key]["algorithmic_bytes"] < 1.15, key
    for key in ("hover65536_240hz:rollout64", "hover65536_240hz:graph"):
        assert c[key]["slots_per_wave_env_step"] > c[key]["valu_per_wave_env_step"] > 0, key
    for wl in ("swarm65536_ext_240hz", "swarm1m_ext_240hz"):
        assert s[wl]["valu_wave_instructions_per_substep"] > 0 and s[wl]["replay_waves"] > 0, wl
