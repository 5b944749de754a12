#!/usr/bin/env python3
"""Copy what round 5 measured on the GPU box (gpurun_out/) into profiles/ (tracked): run in the repo after the gpurun call.

  hbm     gpurun_out/hbm_r05/summary.json (scratch/hbm_reconcile_r05.py) -> profiles/r05_hbm_reconcile.json (+ the by-grid CSVs), the
          file bench.py quotes as `hbm_saturating.rocprof_kernel_avg_us`
  ab      gpurun_out/ab_step_r05*.json / .log -> profiles/r05_ab_step_kernel.*
"""
import json
import os
import shutil
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, P = os.path.join(R, "gpurun_out"), os.path.join(R, "profiles")
what = set(sys.argv[1:]) or {"hbm", "ab", "counters", "bench"}

if "hbm" in what and os.path.exists(os.path.join(G, "hbm_r05", "summary.json")):
    s = json.load(open(os.path.join(G, "hbm_r05", "summary.json")))
    legs = s["legs"]

    def leg(k):
        v = legs[k]
        b = v.get("brief") or {}
        return {"launch_us_hip_events": b.get("launch_us_hip_events"), "frac_hip_events": b.get("frac"), "timed_region_ms": b.get("timed_region_ms"),
                "launches_timed": b.get("launches_timed"), "segments": b.get("segments"), "shader_clock_ghz_after": b.get("clock_ghz_after"),
                "parity_max": b.get("parity_max"), "smi_before": v.get("smi_before"), "smi_after": v.get("smi_after"), "smi_during": v.get("smi_during"),
                "wall_s": v.get("wall_s")}

    tr = s["by_grid_traced"][0]
    bytes_launch = legs["a_plain"]["brief"]["bytes_per_launch"]
    fetch, write = legs["c_pmc_FETCH_SIZE"], legs["c_pmc_WRITE_SIZE"]
    # MI355X_MICROARCH.md, HBM / rocprofv3 section: FETCH_SIZE and WRITE_SIZE count 64-byte... units are KiB-like 1 KB blocks in rocprofv3's
    # derived counters on this stack; gfx950 under-reports FETCH_SIZE by a factor of two (the guide's correction)
    fetch_b = fetch["counter_mean_per_dispatch"] * 1024 * 2
    write_b = write["counter_mean_per_dispatch"] * 1024
    out = {
        "what": "ONE lease, one script (scratch/hbm_reconcile_r05.py): gpd_rollout1_kernel at 4 194 304 drones, 64 env steps per launch, timed plain / "
                "under rocprofv3 --kernel-trace --stats / under --pmc / plain again, rocm-smi clocks around and during every leg; control: a 1 GiB "
                "device-to-device copy plain and traced",
        "bytes_per_launch_algorithmic": bytes_launch,
        "legs": {"a_plain": leg("a_plain"), "b_traced": leg("b_traced"), "c_pmc_FETCH_SIZE": leg("c_pmc_FETCH_SIZE"),
                 "c_pmc_WRITE_SIZE": leg("c_pmc_WRITE_SIZE"), "d_plain": leg("d_plain")},
        "rocprof_by_grid_traced": tr,
        "copy_probe": {"plain_gbs": (legs["a_copy_plain"].get("line") or {}).get("gbs"), "traced_gbs": (legs["b_copy_traced"].get("line") or {}).get("gbs"),
                       "traced_kernel": s.get("copy_by_grid_traced", [{}])[0]},
        "pmc": {"FETCH_SIZE_mean_per_dispatch": fetch["counter_mean_per_dispatch"], "WRITE_SIZE_mean_per_dispatch": write["counter_mean_per_dispatch"],
                "hbm_bytes_per_launch": fetch_b + write_b, "traffic_over_algorithmic": (fetch_b + write_b) / bytes_launch,
                "kernel_avg_us_in_pmc_passes": [fetch.get("kernel_avg_ns_in_pmc_pass", 0) / 1e3, write.get("kernel_avg_ns_in_pmc_pass", 0) / 1e3],
                "note": "FETCH_SIZE x 1024 x 2 (gfx950 correction) + WRITE_SIZE x 1024 bytes per dispatch, separate --pmc passes"},
        "keys": {"hover4m_240hz:rollout64": {
            "rocprof_kernel_avg_us": tr["AverageNs"] / 1e3, "rocprof_calls": tr["Calls"],
            "rocprof_frac": bytes_launch / (tr["AverageNs"] * 1e-9) / 1e9 / 8000.0,
            "hip_events_us_in_the_traced_process": legs["b_traced"]["brief"]["launch_us_hip_events"],
            "hip_events_us_plain_before_after": [legs["a_plain"]["brief"]["launch_us_hip_events"], legs["d_plain"]["brief"]["launch_us_hip_events"]],
            "note": "measured in ONE lease (the reconciliation): there the tracer's durations, the traced process's HIP events and the plain runs agree "
                    "within 2 % -- the tracer does not slow this kernel.  The line quoting this figure ran with ANOTHER allocation of its buffers: this "
                    "leg's rate follows their physical placement (0.62 / 0.68 / 0.72 / 0.75 of 8 TB/s in one process at identical virtual addresses, "
                    "profiles/r05_hbm_placement_modes.txt), so `launch_us_hip_events` of a bench line and this figure may differ by the placement, not "
                    "by the clock (round 4's 3 862 us trace vs 3 147 us bench line were two placements); `on_fresh_allocations` shows the spread"}},
    }
    fr = [out["legs"][k]["frac_hip_events"] for k in ("a_plain", "b_traced", "d_plain")] + [out["keys"]["hover4m_240hz:rollout64"]["rocprof_frac"]]
    out["one_fraction"] = {"value": sum(fr) / len(fr), "min": min(fr), "max": max(fr),
                           "of": "8 TB/s; plain, traced (HIP events and the tracer's own durations) and plain again, all >= 2 s timed regions on one box"}
    json.dump(out, open(os.path.join(P, "r05_hbm_reconcile.json"), "w"), indent=1)
    # the traffic table bench.py reads its `roofline.traffic` / `rocprof_kernel_avg_us` from: this key re-measured in round 5
    tp = os.path.join(P, "hbm_traffic.json")
    t = json.load(open(tp))
    t["hover4m_240hz:rollout64"].update(measured_in_round=5, FETCH_SIZE_KB=fetch["counter_mean_per_dispatch"], WRITE_SIZE_KB=write["counter_mean_per_dispatch"],
                                        traffic_bytes=fetch_b + write_b, rocprof_kernel_avg_ns=tr["AverageNs"],
                                        kernel_avg_ns_in_pmc_pass=fetch.get("kernel_avg_ns_in_pmc_pass"), algorithmic_bytes=int(bytes_launch),
                                        source="profiles/r05_hbm_reconcile.json (one lease: plain / traced / pmc / plain)")
    json.dump(t, open(tp, "w"), indent=1)
    for f in ("b_traced_kernel_stats_by_grid.csv", "b_copy_traced_kernel_stats_by_grid.csv", "c_pmc_FETCH_SIZE_kernel_stats_by_grid.csv"):
        src = os.path.join(G, "hbm_r05", f)
        if os.path.exists(src):
            shutil.copy(src, os.path.join(P, "r05_hbm_" + f))
    print("profiles/r05_hbm_reconcile.json:", json.dumps(out["one_fraction"]), json.dumps(out["pmc"])[:300])

if "ab" in what:
    for f in os.listdir(G):
        if f.startswith("ab_step_r05"):
            shutil.copy(os.path.join(G, f), os.path.join(P, "r05_" + f))
            print("profiles/r05_" + f)

if "counters" in what and os.path.exists(os.path.join(G, "prof_r05", "summary.json")):
    # gpurun_out/prof_r05/summary.json (scratch/profile_r05.py) -> profiles/r05_trace_* / r05_pmc_* + the three tables bench.py reads:
    # the keys measured this round are REPLACED (measured_in_round: 5), the others stay what round 4 measured (their kernels changed in
    # ABI 9 only in the once-per-launch load / store of the state: the per-step figures of the rollout keys are unaffected)
    import glob
    sys.path.insert(0, R)
    SRC = os.path.join(G, "prof_r05")
    d = json.load(open(os.path.join(SRC, "summary.json")))
    for f in glob.glob(os.path.join(P, "r05_pmc_*")) + glob.glob(os.path.join(P, "r05_trace_*")):
        os.remove(f)
    for f in glob.glob(os.path.join(SRC, "*_kernel_stats*.csv")) + glob.glob(os.path.join(SRC, "pmc_*.csv")):
        shutil.copy(f, os.path.join(P, "r05_" + os.path.basename(f)))
    shutil.copy(os.path.join(SRC, "summary.json"), os.path.join(P, "r05_summary.json"))
    for tag, line in d.get("bench_lines", {}).items():
        if line:
            json.dump(line, open(os.path.join(P, f"r05_bench_{tag.replace('trace_', '')}_under_rocprofv3.json"), "w"))
    from bench import WORKLOADS

    def grid_row(tag, frag, grid):
        for r in d["by_grid"].get(tag, []):
            if frag in r["Name"] and int(r["Grid_Size_X"]) == grid:
                return r
        return None

    def alg_bytes(key, spl):
        w = WORKLOADS[key.split(":")[0]]
        N, E = w["E"] * w["D"], w["E"]
        A = {"rpm": 4, "pid": 3, "raw_rpm": 4}[w["act"]]
        pid, drag = w["act"] == "pid", bool(w["phys"] & 2)
        if "rollout" in key:
            state = 2 * 13 * 4 + (2 * 9 * 4 if pid else 0) + (2 * 16 if drag else 0)
            return (state + spl * A * 4 + spl * 48) * N + (8 + spl * 6) * E
        per = (13 + A) * 4 + 25 * 4 + (72 if pid else 0) + (32 if drag else 0)
        return per * N + 14 * E

    traffic = json.load(open(os.path.join(P, "hbm_traffic.json")))
    counters = json.load(open(os.path.join(P, "kernel_counters.json")))
    trace_of = {"hover65536_240hz": "trace_default", "stack8x8192_ext_240hz": "trace_stack8"}
    for key, rec in d.get("pmc", {}).items():
        g = lambda c: rec.get(c, {}).get("mean_per_dispatch")  # noqa: E731
        wl, spl = key.split(":")[0], rec["env_steps_per_launch"]
        if g("FETCH_SIZE") is None or g("WRITE_SIZE") is None:
            print("no traffic counters for", key)
            continue
        t = {"env_steps_per_launch": spl, "algorithmic_bytes": alg_bytes(key, spl), "measured_in_round": 5, "FETCH_SIZE_KB": g("FETCH_SIZE"),
             "WRITE_SIZE_KB": g("WRITE_SIZE"), "traffic_bytes": 2 * g("FETCH_SIZE") * 1024 + g("WRITE_SIZE") * 1024}
        w = WORKLOADS[wl]
        row = grid_row(trace_of.get(wl, ""), rec["kernel"], -(-w["E"] * w["D"] // 256) * 256) if trace_of.get(wl) else None
        if row and ("rollout" in key and spl == 64 or "graph" in key):
            t["rocprof_kernel_avg_ns"] = float(row["AverageNs"])
        if rec.get("kernel_avg_ns_in_pmc_pass"):
            t["kernel_avg_ns_in_pmc_pass"] = rec["kernel_avg_ns_in_pmc_pass"]
        traffic[key] = t
        if g("SQ_INSTS_VALU") is not None and g("SQ_WAVES"):
            per = lambda c: (g(c) or 0.0) / g("SQ_WAVES") / spl  # noqa: E731
            slots = sum(per(c) for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_SMEM"))
            counters[key] = {"slots_per_wave_env_step": slots, "valu_per_wave_env_step": per("SQ_INSTS_VALU"),
                             "salu_per_wave_env_step": per("SQ_INSTS_SALU"), "branch_per_wave_env_step": per("SQ_INSTS_BRANCH"),
                             "lds_per_wave_env_step": per("SQ_INSTS_LDS"), "vmem_per_wave_env_step": per("SQ_INSTS_VMEM_RD") + per("SQ_INSTS_VMEM_WR"),
                             "wave_quad_cycles_per_env_step": per("SQ_WAVE_CYCLES"), "active_quad_cycles_per_env_step": per("SQ_ACTIVE_INST_ANY"),
                             "parked_quad_cycles_per_env_step": per("SQ_WAIT_ANY"), "issue_stall_quad_cycles_per_env_step": per("SQ_WAIT_INST_ANY"),
                             "waves": g("SQ_WAVES"), "measured_in_round": 5}
            print(key, {k: round(v, 1) for k, v in counters[key].items() if isinstance(v, float)})
    traffic["round"], counters["round"] = "4, keys marked measured_in_round 5 re-measured in round 5", "4, keys marked measured_in_round 5 re-measured in round 5"
    json.dump(traffic, open(os.path.join(P, "hbm_traffic.json"), "w"), indent=1)
    json.dump(counters, open(os.path.join(P, "kernel_counters.json"), "w"), indent=1)
    # the one-world kernels
    if d.get("swarm"):
        swarm = {"_comment": "Wave-instructions per physics sub-step of the one-world path (rocprofv3 --pmc SQ_INSTS_*, eager launches, scratch/profile_r05.py, "
                 "the FINAL round-5 library: pair-balanced groups, kernarg preload): per dispatch means of every kernel, and per sub-step = step + replay x "
                 "(1 - 1/rebin) + (build + count + scatter + balance) / rebin with the workload's rebin_every = 16. bench.py prices the VALU instructions x 4 "
                 "cycles on 1024 SIMDs against the measured sub-step (roofline.bound = valu_issue).", "round": 5}
        for wl, rec in d["swarm"].items():
            def get(kfrag, c):
                for kn, cs in rec.items():
                    if kfrag in kn and c in cs:
                        return cs[c]["mean_per_dispatch"]
                return 0.0
            rebin = 16.0
            per_kernel = {kn: {c: v["mean_per_dispatch"] for c, v in cs.items()} for kn, cs in rec.items()}
            if not per_kernel:
                continue

            def per_substep(c):
                return (get("gpd_swarm_step_kernel", c) + get("dwg_force_kernel<2>", c) * (1 - 1 / rebin) +
                        (get("dwg_force_kernel<1>", c) + get("dwg_count_kernel", c) + get("dwg_scatter_kernel", c) + get("dwg_balance_kernel", c)) / rebin)
            swarm[wl] = {"valu_wave_instructions_per_substep": per_substep("SQ_INSTS_VALU"), "salu_wave_instructions_per_substep": per_substep("SQ_INSTS_SALU"),
                         "lds_wave_instructions_per_substep": per_substep("SQ_INSTS_LDS"),
                         "vmem_wave_instructions_per_substep": per_substep("SQ_INSTS_VMEM_RD") + per_substep("SQ_INSTS_VMEM_WR"),
                         "replay_valu_wave_instructions": get("dwg_force_kernel<2>", "SQ_INSTS_VALU"), "replay_waves": get("dwg_force_kernel<2>", "SQ_WAVES"),
                         "replay_wave_quad_cycles": get("dwg_force_kernel<2>", "SQ_WAVE_CYCLES"), "replay_parked_quad_cycles": get("dwg_force_kernel<2>", "SQ_WAIT_ANY"),
                         "replay_active_quad_cycles": get("dwg_force_kernel<2>", "SQ_ACTIVE_INST_ANY"),
                         "hbm_bytes_per_substep": 1024 * (2 * per_substep("FETCH_SIZE") + per_substep("WRITE_SIZE")), "rebin_every": rebin,
                         "per_kernel_per_dispatch": per_kernel,
                         "kernel_avg_ns": {r["Name"].split("::")[-1][:40]: float(r["AverageNs"]) for r in d["traces"].get("trace_" + wl, [])}}
            print(wl, {k: (round(v, 1) if isinstance(v, float) else v) for k, v in swarm[wl].items() if not isinstance(v, dict)})
        json.dump(swarm, open(os.path.join(P, "swarm_counters.json"), "w"), indent=1)

if "bench" in what and os.path.isdir(os.path.join(G, "bench_r05")):
    for f in sorted(os.listdir(os.path.join(G, "bench_r05"))):
        if f.endswith(".json"):
            txt = open(os.path.join(G, "bench_r05", f)).read().strip().splitlines()
            line = next((l for l in reversed(txt) if l.startswith("{")), None)
            if line:
                open(os.path.join(P, "r05_bench_" + f), "w").write(line + "\n")
                print("profiles/r05_bench_" + f)
