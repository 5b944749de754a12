#!/usr/bin/env python3
"""Round-3 profiles, run ON THE GPU BOX (python scratch/profile_r03.py [quick]):

  * rocprofv3 --kernel-trace --stats of the EXACT command the driver times (`bench.py --steps 20 --warmup 5`) and of
    the default command -> average kernel durations that the live `roofline.launch_us_hip_events` must agree with;
  * HBM traffic in separate --pmc passes (FETCH_SIZE, WRITE_SIZE; gfx950: FETCH_SIZE x 2, MI355X_MICROARCH.md);
  * SQ instruction counters (two passes of <= 8 counters) -> instructions per wave and env step for the VALU-issue
    roofline (`profiles/kernel_counters.json`).
Output: gpurun_out/prof_r03/ (+ summary.json).  `scratch/refresh_profiles_r03.py` (run in the repo afterwards) copies
what is to be judged into profiles/."""
import collections
import csv
import glob
import json
import os
import subprocess
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(R, "gpurun_out", "prof_r03")
os.makedirs(OUT, exist_ok=True)
ENV = dict(os.environ, TMPDIR="/tmp")
QUICK = len(sys.argv) > 1 and sys.argv[1] == "quick"

SQ1 = ["SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_SMEM", "SQ_INSTS_BRANCH"]
SQ2 = ["SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_INSTS_VALU"]


def rocprof(tag, prof_args, bench_args, timeout=300):
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


def kernel_stats(tag):
    for f in glob.glob(os.path.join(OUT, tag, "**", "*kernel_stats.csv"), recursive=True):
        rows = [r for r in csv.DictReader(open(f)) if "gpd_" in r["Name"] or "dwg_" in r["Name"]]
        with open(os.path.join(OUT, tag + "_kernel_stats.csv"), "w", newline="") as g:
            wr = csv.DictWriter(g, fieldnames=list(rows[0].keys()) if rows else ["Name"])
            wr.writeheader()
            wr.writerows(rows)
        return rows
    return []


def counters(tag, kern):
    agg = collections.defaultdict(list)
    for f in glob.glob(os.path.join(OUT, tag, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if kern in row["Kernel_Name"]:
                agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
    out = {}
    for k, v in agg.items():
        v = v[len(v) // 4:]                      # (skip the warm-up quarter)
        out[k] = {"mean_per_dispatch": sum(v) / max(len(v), 1), "n": len(v)}
    return out


summary = {"traces": {}, "pmc": {}, "bench_lines": {}}
# ---- kernel traces ------------------------------------------------------------------------------------------------
TRACES = [("trace_driver_cmd", ["--steps", "20", "--warmup", "5", "--no-cpu-baseline"]),
          ("trace_default", ["--no-cpu-baseline"]),
          ("trace_hover65536_240hz_termobs", ["--workload", "hover65536_240hz_termobs", "--no-cpu-baseline"]),
          ("trace_swarm65536", ["--workload", "swarm65536_ext_240hz", "--steps", "240", "--warmup", "24", "--no-cpu-baseline"])]
if not QUICK:
    TRACES += [("trace_hover65536_30hz", ["--workload", "hover65536_30hz", "--no-cpu-baseline"]),
               ("trace_hover4m", ["--workload", "hover4m_240hz", "--no-cpu-baseline", "--steps", "256", "--warmup", "64"])]
for tag, args in TRACES:
    line = rocprof(tag, ["--kernel-trace", "--stats"], args)
    summary["traces"][tag] = kernel_stats(tag)
    summary["bench_lines"][tag] = line

# ---- counters -----------------------------------------------------------------------------------------------------
PMC = [  # key, workload, mode, kernel name fragment, steps per launch
    ("hover65536_240hz:rollout64", "hover65536_240hz", "rollout", "gpd_rollout", 64),
    ("hover65536_240hz:graph", "hover65536_240hz", "eager", "gpd_step_kernel", 1),
]
if not QUICK:
    PMC += [("hover65536_30hz:rollout64", "hover65536_30hz", "rollout", "gpd_rollout", 64),
            ("hover65536_240hz_termobs:rollout64", "hover65536_240hz_termobs", "rollout", "gpd_rollout", 64),
            ("hover65536_pid_240hz:rollout64", "hover65536_pid_240hz", "rollout", "gpd_rollout", 64),
            ("multihover2x16384_240hz:rollout64", "multihover2x16384_240hz", "rollout", "gpd_rollout", 64)]
for key, wl, mode, kern, spl in PMC:
    rec = {"env_steps_per_launch": spl, "kernel": kern}
    big = wl in ("hover4m_240hz", "hover16m_240hz")
    groups = [("FETCH_SIZE", ["FETCH_SIZE"]), ("WRITE_SIZE", ["WRITE_SIZE"])] + ([] if big else [("SQ1", SQ1), ("SQ2", SQ2)])
    for gname, ctrs in groups:
        tag = "pmc_" + key.replace(":", "_") + "_" + gname
        args = ["--workload", wl, "--mode", mode, "--no-cpu-baseline", "--no-second-leg", "--steps", "64", "--warmup", "64",
                "--min-time", "0.002" if not big else "0.0001"]
        rocprof(tag, ["--kernel-trace", "--pmc"] + ctrs, args)
        for c, v in counters(tag, kern).items():
            rec.setdefault(c, v)
        f = next(iter(glob.glob(os.path.join(OUT, tag, "**", "*counter_collection.csv"), recursive=True)), None)
        if f:   # a readable excerpt of the raw rows for profiles/
            lines = open(f).read().splitlines()
            keep = [lines[0]] + [l for l in lines[1:] if kern in l][-60:]
            open(os.path.join(OUT, tag + ".csv"), "w").write("\n".join(keep) + "\n")
    summary["pmc"][key] = rec
json.dump(summary, open(os.path.join(OUT, "summary.json"), "w"), indent=1)
# the raw per-dispatch traces are tens of MB per pass (gpurun copies back 64 MiB at most): keep the extracted summaries only
import shutil
for d in glob.glob(os.path.join(OUT, "*")):
    if os.path.isdir(d):
        shutil.rmtree(d, ignore_errors=True)
for key, rec in summary["pmc"].items():
    print(key, {k: (round(v["mean_per_dispatch"], 1) if isinstance(v, dict) else v) for k, v in rec.items()})
for tag, rows in summary["traces"].items():
    for r in rows:
        print(tag, r["Name"][:60], "calls", r.get("Calls"), "avg ns", r.get("AverageNs"))
