#!/usr/bin/env python3
"""Round-5 profiles (the round-4 script with this round's key list), run ON THE GPU BOX (python scratch/profile_r05.py [traces] [pmc] [swarm]):

  traces  rocprofv3 --kernel-trace --stats of the EXACT command the driver times (`bench.py --steps 20 --warmup 5`, which now
          also runs the HBM-saturating leg: the same kernel at 4 194 304 drones) and of the default command.  The stats CSV of
          rocprofv3 averages over both batch sizes, so the raw per-dispatch trace is ALSO split by grid size here
          (`*_by_grid.csv`): the 65 536-drone rows are what `roofline.launch_us_hip_events` must agree with, the 4 194 304-drone
          rows what `hbm_saturating` must.  Plus the one-launch-per-step graph leg dispatch by dispatch: duration, period and gap
          of consecutive gpd_step_kernel launches -- the reconciliation of rocprof's 4.97 us average kernel duration with the
          4.00 us per step the HIP events see.
  pmc     HBM traffic (FETCH_SIZE, WRITE_SIZE; separate passes; gfx950: FETCH_SIZE x 2) and the SQ instruction counters for EVERY
          key of profiles/hbm_traffic.json / kernel_counters.json -- no entry of round 2 or 3 survives.
  swarm   the one-world kernels: trace + SQ counters of swarm65536 / swarm1m (pairs, instructions per pair: bench.py's pair
          roofline reads profiles/swarm_counters.json).
Output: gpurun_out/prof_r05/ (+ summary.json).  `scratch/refresh_profiles_r05.py counters` (run in the repo afterwards) copies what is to
be judged into profiles/."""
import collections
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(R, "gpurun_out", "prof_r05")
os.makedirs(OUT, exist_ok=True)
ENV = dict(os.environ, TMPDIR="/tmp")
WHAT = set(sys.argv[1:]) or {"traces", "pmc", "swarm"}

SQ1 = ["SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_SMEM", "SQ_INSTS_BRANCH"]
SQ2 = ["SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_INSTS_VALU"]


def rocprof(tag, prof_args, bench_args, timeout=420):
    d = os.path.join(OUT, tag)
    subprocess.run(["rm", "-rf", d])
    cmd = ["rocprofv3"] + prof_args + ["--output-format", "csv", "-d", d, "-o", "p", "--", sys.executable, os.path.join(R, "bench.py")] + bench_args
    try:
        res = subprocess.run(cmd, cwd="/tmp", env=ENV, capture_output=True, text=True, timeout=timeout)
    except subprocess.TimeoutExpired:
        print(tag, "TIMEOUT", flush=True)
        return None
    open(os.path.join(OUT, tag + ".log"), "w").write(res.stdout[-20000:] + "\n---- stderr ----\n" + res.stderr[-5000:])
    line = next((l for l in reversed(res.stdout.splitlines()) if l.startswith("{")), None)
    print(tag, "rc", res.returncode, flush=True)
    return json.loads(line) if line else None


def find(tag, pattern):
    return next(iter(glob.glob(os.path.join(OUT, tag, "**", pattern), recursive=True)), None)


def kernel_stats(tag):
    f = find(tag, "*kernel_stats.csv")
    if not f:
        return []
    rows = [r for r in csv.DictReader(open(f)) if "gpd_" in r["Name"] or "dwg_" in r["Name"]]
    with open(os.path.join(OUT, tag + "_kernel_stats.csv"), "w", newline="") as g:
        wr = csv.DictWriter(g, fieldnames=list(rows[0].keys()) if rows else ["Name"])
        wr.writeheader()
        wr.writerows(rows)
    return rows


def dispatches(tag):
    """the raw trace: (kernel name, start ns, end ns, grid size x) per dispatch, in start order"""
    f = find(tag, "*kernel_trace.csv")
    if not f:
        return []
    out = []
    for r in csv.DictReader(open(f)):
        name = r.get("Kernel_Name", "")
        if "gpd_" not in name and "dwg_" not in name:
            continue
        out.append((name, int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0)))
    out.sort(key=lambda t: t[1])
    return out


def by_grid(tag):
    agg = collections.defaultdict(list)
    for name, s, e, g in dispatches(tag):
        agg[(name, g)].append(e - s)
    rows = [{"Name": n, "Grid_Size_X": g, "Calls": len(v), "TotalNs": sum(v), "AverageNs": sum(v) / len(v), "MinNs": min(v), "MaxNs": max(v)}
            for (n, g), v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))]
    with open(os.path.join(OUT, tag + "_kernel_stats_by_grid.csv"), "w", newline="") as f:
        wr = csv.DictWriter(f, fieldnames=["Name", "Grid_Size_X", "Calls", "TotalNs", "AverageNs", "MinNs", "MaxNs"])
        wr.writeheader()
        wr.writerows(rows)
    return rows


def counters(tag, kern):
    agg = collections.defaultdict(list)
    f = find(tag, "*counter_collection.csv")
    if f:
        for row in csv.DictReader(open(f)):
            if kern in row["Kernel_Name"]:
                agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
        lines = open(f).read().splitlines()
        keep = [lines[0]] + [l for l in lines[1:] if kern in l][-60:]
        open(os.path.join(OUT, tag + ".csv"), "w").write("\n".join(keep) + "\n")
    out = {}
    for k, v in agg.items():
        v = v[len(v) // 4:]                      # (skip the warm-up quarter)
        out[k] = {"mean_per_dispatch": sum(v) / max(len(v), 1), "n": len(v)}
    return out


summary = {"traces": {}, "by_grid": {}, "pmc": {}, "bench_lines": {}, "graph_leg": {}, "swarm": {}}
# a partial run (e.g. `swarm` after a change of the one-world kernels only) starts from the committed summary: the sections it does
# not measure stay what they were
_prev = os.path.join(R, "profiles", "r05_summary.json")
if WHAT != {"traces", "pmc", "swarm"} and os.path.exists(_prev):
    _old = json.load(open(_prev))
    for _k in summary:
        summary[_k] = _old.get(_k, summary[_k])
if os.path.exists(os.path.join(OUT, "summary.json")):
    try:
        summary.update(json.load(open(os.path.join(OUT, "summary.json"))))
    except Exception:
        pass

if "traces" in WHAT:
    TRACES = [("trace_driver_cmd", ["--steps", "20", "--warmup", "5", "--no-cpu-baseline"]),
              ("trace_default", ["--no-cpu-baseline"]),
              ("trace_stack8", ["--workload", "stack8x8192_ext_240hz", "--no-cpu-baseline"])]
    for tag, args in TRACES:
        line = rocprof(tag, ["--kernel-trace", "--stats"], args)
        summary["traces"][tag] = kernel_stats(tag)
        summary["by_grid"][tag] = by_grid(tag)
        summary["bench_lines"][tag] = line
    # the graph leg, dispatch by dispatch: one hipGraph of 64 gpd_step launches replayed back to back
    tag = "trace_graph_leg"
    line = rocprof(tag, ["--kernel-trace"], ["--mode", "graph", "--steps", "64", "--warmup", "8", "--no-cpu-baseline", "--no-hbm-leg", "--no-parity"])
    d = [x for x in dispatches(tag) if "gpd_step_kernel" in x[0]]
    d = d[len(d) // 4:]
    dur = [e - s for _, s, e, _ in d]
    period = [d[i + 1][1] - d[i][1] for i in range(len(d) - 1)]
    gap = [d[i + 1][1] - d[i][2] for i in range(len(d) - 1)]
    inside = [(p, g) for p, g in zip(period, gap) if p < 20000]          # consecutive launches of one graph replay (a replay boundary is longer)
    rec = {"dispatches": len(d), "kernel_duration_avg_ns": sum(dur) / max(len(dur), 1),
           "start_to_start_avg_ns": sum(p for p, _ in inside) / max(len(inside), 1),
           "end_to_next_start_avg_ns": sum(g for _, g in inside) / max(len(inside), 1),
           "overlapping_pairs_frac": sum(1 for _, g in inside if g < 0) / max(len(inside), 1),
           "bench_us_per_step_hip_events": (line or {}).get("ms_per_step", 0) * 1e3,
           "first_rows_ns": [[s - d[0][1], e - d[0][1]] for _, s, e, _ in d[:12]]}
    summary["graph_leg"] = rec
    print("graph leg:", {k: v for k, v in rec.items() if k != "first_rows_ns"})

if "pmc" in WHAT:
    PMC = [  # key, workload, mode, kernel name fragment, steps per launch, extra args
        ("hover65536_240hz:rollout64", "hover65536_240hz", "rollout", "gpd_rollout", 64),
        ("hover65536_240hz:graph", "hover65536_240hz", "eager", "gpd_step_kernel", 1),
        ("hover65536_pid_240hz:graph", "hover65536_pid_240hz", "eager", "gpd_step_kernel", 1),
        ("hover65536_30hz:graph", "hover65536_30hz", "eager", "gpd_step_kernel", 1),
        ("hover16m_240hz:graph", "hover16m_240hz", "eager", "gpd_step_kernel", 1),
    ]
    for key, wl, mode, kern, spl in PMC:
        rec = {"env_steps_per_launch": spl, "kernel": kern}
        big = wl in ("hover4m_240hz", "hover16m_240hz")
        groups = [("FETCH_SIZE", ["FETCH_SIZE"]), ("WRITE_SIZE", ["WRITE_SIZE"])] + ([] if big else [("SQ1", SQ1), ("SQ2", SQ2)])
        for gname, ctrs in groups:
            tag = "pmc_" + key.replace(":", "_") + "_" + gname
            args = ["--workload", wl, "--mode", mode, "--no-cpu-baseline", "--no-second-leg", "--no-hbm-leg", "--no-parity", "--steps", str(spl if spl > 1 else 64),
                    "--warmup", str(spl if spl > 1 else 64), "--min-time", "0.002" if not big else "0.0001"]
            rocprof(tag, ["--kernel-trace", "--pmc"] + ctrs, args)
            for c, v in counters(tag, kern).items():
                rec.setdefault(c, v)
            if gname == "FETCH_SIZE":           # the kernel's duration in that very pass, for the record
                dd = [e - s for n, s, e, g in dispatches(tag) if kern in n]
                dd = dd[len(dd) // 4:]
                if dd:
                    rec["kernel_avg_ns_in_pmc_pass"] = sum(dd) / len(dd)
        summary["pmc"][key] = rec
        print(key, {k: (round(v["mean_per_dispatch"], 1) if isinstance(v, dict) else v) for k, v in rec.items()}, flush=True)

if "swarm" in WHAT:
    for wl, steps in (("swarm65536_ext_240hz", 240), ("swarm1m_ext_240hz", 64)):
        tag = "trace_" + wl
        line = rocprof(tag, ["--kernel-trace", "--stats"], ["--workload", wl, "--steps", str(steps), "--warmup", "24", "--no-cpu-baseline"], timeout=600)
        summary["traces"][tag] = kernel_stats(tag)
        summary["bench_lines"][tag] = line
        rec = {}
        for gname, ctrs in (("SQ1", SQ1), ("SQ2", SQ2), ("FETCH_SIZE", ["FETCH_SIZE"]), ("WRITE_SIZE", ["WRITE_SIZE"])):
            t2 = "pmc_" + wl + "_" + gname
            rocprof(t2, ["--kernel-trace", "--pmc"] + ctrs, ["--workload", wl, "--mode", "eager", "--steps", "64", "--warmup", "16", "--min-time", "0.002",
                                                             "--no-cpu-baseline", "--no-parity"], timeout=600)
            # per kernel (the force kernel by template argument: <1> sweep + build lists, <2> replay lists, <0> no lists)
            f = find(t2, "*counter_collection.csv")
            if f:
                import re
                per = collections.defaultdict(lambda: collections.defaultdict(list))
                for row in csv.DictReader(open(f)):
                    mm = re.search(r"(dwg_\w+|gpd_swarm_\w+)(<\w+>)?", row["Kernel_Name"])
                    if mm:
                        per[mm.group(0)][row["Counter_Name"]].append(float(row["Counter_Value"]))
                for kn, cs in per.items():
                    for c, v in cs.items():
                        rec.setdefault(kn, {}).setdefault(c, {"mean_per_dispatch": sum(v) / len(v), "n": len(v)})
                lines = open(f).read().splitlines()
                keep = [lines[0]] + [l for l in lines[1:] if "dwg_" in l or "gpd_swarm" in l][-80:]
                open(os.path.join(OUT, t2 + ".csv"), "w").write("\n".join(keep) + "\n")
        summary["swarm"][wl] = rec
        print(wl, json.dumps(rec)[:600], flush=True)

json.dump(summary, open(os.path.join(OUT, "summary.json"), "w"), indent=1)
# the raw per-dispatch traces are tens of MB per pass (gpurun copies back 64 MiB at most): keep the extracted summaries only
for d in glob.glob(os.path.join(OUT, "*")):
    if os.path.isdir(d):
        shutil.rmtree(d, ignore_errors=True)
for tag, rows in summary["by_grid"].items():
    for r in rows[:6]:
        print(tag, r["Name"][:70], "grid", r["Grid_Size_X"], "calls", r["Calls"], "avg ns %.0f" % r["AverageNs"])
