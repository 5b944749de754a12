#!/usr/bin/env python3
"""Copy the judged summaries of gpurun_out/prof_r03 (scratch/profile_r03.py) and the bench JSON lines into profiles/ and
rebuild profiles/hbm_traffic.json + profiles/kernel_counters.json (what bench.py attaches to its live line)."""
import glob
import json
import os
import shutil
import sys

sys.path.insert(0, ".")
SRC = "gpurun_out/prof_r03"
d = json.load(open(f"{SRC}/summary.json"))
for f in glob.glob("profiles/r03_pmc_*") + ["profiles/r03_summary.json"]:
    if os.path.exists(f):
        os.remove(f)
for f in glob.glob(f"{SRC}/*_kernel_stats.csv") + glob.glob(f"{SRC}/pmc_*.csv"):
    shutil.copy(f, "profiles/r03_" + os.path.basename(f))
shutil.copy(f"{SRC}/summary.json", "profiles/r03_summary.json")
for f in glob.glob("gpurun_out/r03_bench_*.json"):
    if os.path.getsize(f):
        shutil.copy(f, "profiles/" + os.path.basename(f))
# (only the text files that are copied as they come; the others in profiles/ are curated -- before / after in one file, renamed)
for name in ("r03_smoke_diag.txt", "r03_swarm_partition_cost.txt"):
    if os.path.exists("gpurun_out/" + name):
        shutil.copy("gpurun_out/" + name, "profiles/" + name)


def avg_ns(tag, frag):
    for r in d["traces"].get(tag, []):
        if frag in r["Name"]:
            return float(r["AverageNs"])
    return None


ALG = {}       # algorithmic bytes per profiled launch, from the bench lines the traced commands printed


def alg_bytes(key):
    from bench import WORKLOADS
    w = WORKLOADS[key.split(":")[0]]
    N, E = w["E"] * w["D"], w["E"]
    A = {"rpm": 4, "pid": 3, "raw_rpm": 4}[w["act"]]
    pid = w["act"] == "pid"
    drag = bool(w["phys"] & 2)
    if key.endswith("rollout64"):
        state = 2 * 13 * 4 + (2 * 9 * 4 if pid else 0) + (2 * 16 if drag else 0)
        return (state + 64 * A * 4 + 64 * 48) * N + (8 + 64 * 6) * E
    per = (13 + A) * 4 + 25 * 4 + (72 if pid else 0) + (32 if drag else 0)
    return per * N + 14 * E


traffic = {"_comment": "HBM traffic per kernel dispatch from rocprofv3 --pmc (separate passes for FETCH_SIZE and WRITE_SIZE; "
           "scratch/profile_r03.py). bytes = 2*FETCH_SIZE_KB*1024 + WRITE_SIZE_KB*1024: on gfx950 FETCH_SIZE reports half of the "
           "bytes of a coalesced stream (MI355X_MICROARCH.md, HBM section). Keys are '<bench workload>:<launch mode>'; a rollout "
           "dispatch is 64 env steps, bench.py scales the figure by the algorithmic bytes when it times another step count. "
           "rocprof_kernel_avg_ns: --kernel-trace --stats of the bench command (profiles/r03_trace_*_kernel_stats.csv).",
           "round": 3}
counters = {"_comment": "Instructions per wavefront and env step from rocprofv3 --pmc SQ_INSTS_* (scratch/profile_r03.py): value per "
            "dispatch / SQ_WAVES / env steps per launch. slots = VALU + SALU + LDS + VMEM_RD + VMEM_WR + SMEM (SALU includes "
            "branches, waits and nops). bench.py prices slots x 4 cycles (a wave64 VALU op occupies its SIMD for 4 cycles) against "
            "the measured step time: roofline_valu_issue.", "round": 3}
trace_of = {"hover65536_240hz": "trace_default", "hover65536_30hz": "trace_hover65536_30hz", "hover4m_240hz": "trace_hover4m",
            "hover65536_240hz_termobs": "trace_hover65536_240hz_termobs"}
# (keys this round's passes did not re-measure keep their round-2 record)
try:
    old_t, old_c = json.load(open("profiles/hbm_traffic.json")), json.load(open("profiles/kernel_counters.json"))
except Exception:
    old_t, old_c = {}, {}
for key, rec in d["pmc"].items():
    g = lambda c: rec.get(c, {}).get("mean_per_dispatch")  # noqa: E731
    wl = key.split(":")[0]
    t = {"env_steps_per_launch": rec["env_steps_per_launch"], "algorithmic_bytes": alg_bytes(key)}
    if g("FETCH_SIZE") is not None and g("WRITE_SIZE") is not None:
        t.update(FETCH_SIZE_KB=g("FETCH_SIZE"), WRITE_SIZE_KB=g("WRITE_SIZE"), traffic_bytes=2 * g("FETCH_SIZE") * 1024 + g("WRITE_SIZE") * 1024)
    a = avg_ns(trace_of.get(wl, ""), rec["kernel"])
    if a:
        t["rocprof_kernel_avg_ns"] = a
    traffic[key] = t
    if g("SQ_INSTS_VALU") is not None and g("SQ_WAVES"):
        per = lambda c: (g(c) or 0.0) / g("SQ_WAVES") / rec["env_steps_per_launch"]  # noqa: E731
        slots = sum(per(c) for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_SMEM"))
        counters[key] = {"slots_per_wave_env_step": slots, "valu_per_wave_env_step": per("SQ_INSTS_VALU"),
                         "salu_per_wave_env_step": per("SQ_INSTS_SALU"), "branch_per_wave_env_step": per("SQ_INSTS_BRANCH"),
                         "lds_per_wave_env_step": per("SQ_INSTS_LDS"), "vmem_per_wave_env_step": per("SQ_INSTS_VMEM_RD") + per("SQ_INSTS_VMEM_WR"),
                         "wave_quad_cycles_per_env_step": per("SQ_WAVE_CYCLES"), "active_quad_cycles_per_env_step": per("SQ_ACTIVE_INST_ANY"),
                         "parked_quad_cycles_per_env_step": per("SQ_WAIT_ANY"), "issue_stall_quad_cycles_per_env_step": per("SQ_WAIT_INST_ANY"),
                         "waves": g("SQ_WAVES")}
for k, v in old_t.items():
    if isinstance(v, dict) and k not in traffic:
        traffic[k] = dict(v, measured_in_round=v.get("measured_in_round", 2))
for k, v in old_c.items():
    if isinstance(v, dict) and k not in counters:
        counters[k] = dict(v, measured_in_round=v.get("measured_in_round", 2))
json.dump(traffic, open("profiles/hbm_traffic.json", "w"), indent=1)
json.dump(counters, open("profiles/kernel_counters.json", "w"), indent=1)
for name, tab in (("traffic", traffic), ("counters", counters)):
    for k, v in tab.items():
        if isinstance(v, dict):
            print(name, k, {a: (round(b, 2) if isinstance(b, float) else b) for a, b in v.items()})
