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
