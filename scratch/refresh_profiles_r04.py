#!/usr/bin/env python3
"""Copy the judged summaries of gpurun_out/prof_r04 (scratch/profile_r04.py) into profiles/ and rebuild
profiles/hbm_traffic.json + profiles/kernel_counters.json + profiles/swarm_counters.json (what bench.py attaches to its live
line).  Every key is re-measured in round 4: nothing of rounds 2 / 3 is carried over (an entry that could not be measured is
dropped and listed)."""
import glob
import json
import os
import shutil
import sys

sys.path.insert(0, ".")
SRC = "gpurun_out/prof_r04"
d = json.load(open(f"{SRC}/summary.json"))
for f in glob.glob("profiles/r04_pmc_*") + glob.glob("profiles/r04_trace_*"):
    os.remove(f)
for f in glob.glob(f"{SRC}/*_kernel_stats*.csv") + glob.glob(f"{SRC}/pmc_*.csv"):
    shutil.copy(f, "profiles/r04_" + os.path.basename(f))
shutil.copy(f"{SRC}/summary.json", "profiles/r04_summary.json")
for tag, line in d.get("bench_lines", {}).items():          # the JSON lines the traced commands printed
    if line:
        json.dump(line, open(f"profiles/r04_bench_{tag.replace('trace_', '')}_under_rocprofv3.json", "w"))


def grid_row(tag, frag, grid):
    for r in d["by_grid"].get(tag, []):
        if frag in r["Name"] and int(r["Grid_Size_X"]) == grid:
            return r
    return None


def alg_bytes(key, spl):
    from bench import WORKLOADS
    w = WORKLOADS[key.split(":")[0]]
    N, E = w["E"] * w["D"], w["E"]
    A = {"rpm": 4, "pid": 3, "raw_rpm": 4}[w["act"]]
    pid = w["act"] == "pid"
    drag = bool(w["phys"] & 2)
    tobs = bool(w.get("term_obs"))
    if "rollout" in key:
        state = 2 * 13 * 4 + (2 * 9 * 4 if pid else 0) + (2 * 16 if drag else 0)
        return (state + spl * A * 4 + spl * 48) * N + (8 + spl * 6) * E
    per = (13 + A) * 4 + 25 * 4 + (72 if pid else 0) + (32 if drag else 0)
    return per * N + 14 * E


traffic = {"_comment": "HBM-side traffic per kernel dispatch from rocprofv3 --pmc (separate passes for FETCH_SIZE and WRITE_SIZE; "
           "scratch/profile_r04.py). bytes = 2*FETCH_SIZE_KB*1024 + WRITE_SIZE_KB*1024: on gfx950 FETCH_SIZE reports half of the bytes "
           "of a coalesced stream (MI355X_MICROARCH.md, HBM section). The counters sit between L2 and the fabric: Infinity-Cache hits "
           "are counted -- for the 65 536-drone keys (<= 300 MB per launch, re-used by every launch) this is fabric-side traffic, for "
           "hover4m / hover16m (GBs per launch) it is HBM traffic. Keys are '<bench workload>:<launch mode>'; bench.py scales the "
           "figure by the algorithmic bytes when it times another step count. rocprof_kernel_avg_ns: the kernel's average duration in "
           "the --kernel-trace of the bench command, rows of THIS batch size only (profiles/r04_trace_*_kernel_stats_by_grid.csv).",
           "round": 4}
counters = {"_comment": "Instructions per wavefront and env step from rocprofv3 --pmc SQ_INSTS_* (scratch/profile_r04.py): value per "
            "dispatch / SQ_WAVES / env steps per launch. slots = VALU + SALU + LDS + VMEM_RD + VMEM_WR + SMEM (SALU includes branches, "
            "waits and nops). bench.py prices slots x 4 cycles (a wave64 VALU op occupies its SIMD for 4 cycles) against the measured "
            "step time: roofline_valu_issue.", "round": 4}
trace_of = {"hover65536_240hz": "trace_default", "hover65536_30hz": "trace_hover65536_30hz", "hover4m_240hz": "trace_hover4m",
            "hover65536_240hz_termobs": "trace_hover65536_240hz_termobs", "stack8x8192_ext_240hz": "trace_stack8"}
dropped = []
for key, rec in d["pmc"].items():
    g = lambda c: rec.get(c, {}).get("mean_per_dispatch")  # noqa: E731
    wl = key.split(":")[0]
    spl = rec["env_steps_per_launch"]
    t = {"env_steps_per_launch": spl, "algorithmic_bytes": alg_bytes(key, spl), "measured_in_round": 4}
    if g("FETCH_SIZE") is not None and g("WRITE_SIZE") is not None:
        t.update(FETCH_SIZE_KB=g("FETCH_SIZE"), WRITE_SIZE_KB=g("WRITE_SIZE"), traffic_bytes=2 * g("FETCH_SIZE") * 1024 + g("WRITE_SIZE") * 1024)
    else:
        dropped.append(key)
        continue
    from bench import WORKLOADS
    w = WORKLOADS[wl]
    row = grid_row(trace_of.get(wl, ""), rec["kernel"], -(-w["E"] * w["D"] // 256) * 256) if trace_of.get(wl) else None
    if row and "rollout" in key and spl == 64 or row and "graph" in key:
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
                         "waves": g("SQ_WAVES"), "measured_in_round": 4}
json.dump(traffic, open("profiles/hbm_traffic.json", "w"), indent=1)
json.dump(counters, open("profiles/kernel_counters.json", "w"), indent=1)

# ---- the one-world kernels: instruction counts per sub-step ---------------------------------------------------------------
swarm = {"_comment": "Wave-instructions per physics sub-step of the one-world path (rocprofv3 --pmc SQ_INSTS_*, eager launches, "
         "scratch/profile_r04.py): per dispatch means of every kernel, and per sub-step = step + replay x (1 - 1/rebin) + (build + "
         "count + scatter) / rebin with the workload's rebin_every = 16. bench.py prices the VALU instructions x 4 cycles on 1024 "
         "SIMDs against the measured sub-step (roofline.bound = valu_issue).", "round": 4}
for wl, rec in d.get("swarm", {}).items():
    def get(kfrag, c):
        for kn, cs in rec.items():
            if kfrag in kn and c in cs:
                return cs[c]["mean_per_dispatch"]
        return 0.0
    rebin = 16.0
    per_kernel = {}
    for kn, cs in rec.items():
        per_kernel[kn] = {c: v["mean_per_dispatch"] for c, v in cs.items()}
    def per_substep(c):
        return (get("gpd_swarm_step_kernel", c) + get("dwg_force_kernel<2>", c) * (1 - 1 / rebin) +
                (get("dwg_force_kernel<1>", c) + get("dwg_count_kernel", c) + get("dwg_scatter_kernel", c)) / rebin)
    if not per_kernel:
        continue
    swarm[wl] = {"valu_wave_instructions_per_substep": per_substep("SQ_INSTS_VALU"),
                 "salu_wave_instructions_per_substep": per_substep("SQ_INSTS_SALU"),
                 "lds_wave_instructions_per_substep": per_substep("SQ_INSTS_LDS"),
                 "vmem_wave_instructions_per_substep": per_substep("SQ_INSTS_VMEM_RD") + per_substep("SQ_INSTS_VMEM_WR"),
                 "replay_valu_wave_instructions": get("dwg_force_kernel<2>", "SQ_INSTS_VALU"),
                 "replay_waves": get("dwg_force_kernel<2>", "SQ_WAVES"),
                 "replay_wave_quad_cycles": get("dwg_force_kernel<2>", "SQ_WAVE_CYCLES"),
                 "replay_parked_quad_cycles": get("dwg_force_kernel<2>", "SQ_WAIT_ANY"),
                 "replay_active_quad_cycles": get("dwg_force_kernel<2>", "SQ_ACTIVE_INST_ANY"),
                 "hbm_bytes_per_substep": 1024 * (2 * per_substep("FETCH_SIZE") + per_substep("WRITE_SIZE")),
                 "rebin_every": rebin, "per_kernel_per_dispatch": per_kernel,
                 "kernel_avg_ns": {r["Name"].split("::")[-1][:40]: float(r["AverageNs"]) for r in d["traces"].get("trace_" + wl, [])}}
json.dump(swarm, open("profiles/swarm_counters.json", "w"), indent=1)

# ---- the graph leg: which clock says what -----------------------------------------------------------------------------------
g = d.get("graph_leg") or {}
if g:
    with open("profiles/r04_graph_leg_clocks.txt", "w") as f:
        f.write("One hipGraph of 64 gpd_step launches (hover65536_240hz, `bench.py --mode graph --steps 64`), replayed back to back, under\n"
                "rocprofv3 --kernel-trace: the per-dispatch timestamps of consecutive gpd_step_kernel launches (scratch/profile_r04.py).\n\n")
        for k, v in g.items():
            if k != "first_rows_ns":
                f.write(f"{k}: {v}\n")
        f.write("\nfirst dispatches [start, end] in ns relative to the first start:\n")
        for s_, e_ in g.get("first_rows_ns", []):
            f.write(f"  {s_:8d} .. {e_:8d}   (duration {e_ - s_})\n")
        f.write("\nReading.  VERDICT r03 (weak #3) asked which clock is wrong: profiles/hbm_traffic.json gave `rocprof_kernel_avg_us` 4.97 for\n"
                "gpd_step_kernel in the graph leg, the bench line 4.00 us per step from HIP events -- and a kernel cannot last longer than the\n"
                "period of back-to-back launches.  Neither is wrong; they are not the same run.  Under rocprofv3 --kernel-trace the launch\n"
                "PERIOD itself (`start_to_start`) is 5.3 us and the bench's own HIP events in that very process read\n"
                "`bench_us_per_step_hip_events` (5.6): the tracer's per-dispatch instrumentation costs ~1.3 us per launch at this launch\n"
                "rate (a third of the step), and the kernel's traced duration (5.1 us) fits inside the traced period with a small positive gap\n"
                "(`end_to_next_start`; consecutive intervals do not overlap).  The unprofiled figure (4.00 us per step, HIP events around the\n"
                "graph replays, no tool attached) is the rate; a traced duration is only comparable with traced periods.  For the 20- and\n"
                "64-step rollout launches (20 / 52 us each) the same ~1 us is 2-5 % -- which is the agreement the bench line and the\n"
                "kernel-stats CSV show there.\n")
print("dropped (no counters):", dropped)
for name, tab in (("traffic", traffic), ("counters", counters), ("swarm", swarm)):
    for k, v in tab.items():
        if isinstance(v, dict):
            print(name, k, {a: (round(b, 2) if isinstance(b, float) else b) for a, b in v.items() if not isinstance(b, dict)})
