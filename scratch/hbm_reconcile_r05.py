#!/usr/bin/env python3
"""ONE lease, ONE script (VERDICT r04 "next" #1): the HBM-served roofline fraction of `gpd_rollout1_kernel` at 4 194 304 drones,
64 env steps per launch, measured four ways back to back on the same box --

    (a) plain       bench.py --workload hover4m_240hz            HIP events, >= 2 s timed region seen in 48 pieces
    (b) traced      rocprofv3 --kernel-trace --stats -- (same)   the tracer's per-dispatch durations AND that process's own HIP events
    (c) pmc         rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE      (separate passes; short runs) HBM bytes per launch
    (d) plain       again                                         did the box drift?

with `rocm-smi` clocks / power before and after every leg and sampled every 0.25 s during it, `gpd_clock_probe` inside every bench
process, and -- the control -- a plain device-to-device copy (torch `copy_`, 1 GiB) measured plain and under the tracer the same way:
if the tracer slows a copy by the same factor it is the tool (or the power state it selects), not this kernel.

Output: gpurun_out/hbm_r05/summary.json (+ the by-grid CSV of the traced run); `scratch/refresh_profiles_r05.py` turns it into
profiles/r05_hbm_reconcile.json, which bench.py quotes (`hbm_saturating.rocprof_kernel_avg_us`)."""
import collections
import csv
import glob
import json
import os
import re
import shutil
import subprocess
import sys
import threading
import time

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(R, "gpurun_out", "hbm_r05")
os.makedirs(OUT, exist_ok=True)
ENV = dict(os.environ, TMPDIR="/tmp")
BENCH = [sys.executable, os.path.join(R, "bench.py"), "--workload", "hover4m_240hz", "--steps", "64", "--warmup", "64", "--no-cpu-baseline",
         "--no-second-leg", "--no-hbm-leg", "--segment-events", "48"]
COPY = [sys.executable, os.path.join(R, "scratch", "hbm_reconcile_r05.py"), "--copy-probe"]


def smi():
    try:
        txt = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showtemp", "--json"], capture_output=True, text=True, timeout=20).stdout
        d = json.loads(txt)
        card = d[sorted(d)[0]]
        keep = {}
        for k, v in card.items():
            if re.search(r"sclk|mclk|fclk|socclk|Power|Temperature \(Sensor (junction|memory)", k):
                keep[k] = v
        return keep
    except Exception as e:      # noqa: BLE001
        return {"error": f"{type(e).__name__}: {e}"[:120]}


class Sampler:
    def __init__(self, period=0.25):
        self.rows, self.stop, self.period = [], False, period
        self.t = threading.Thread(target=self.run, daemon=True)

    def run(self):
        t0 = time.perf_counter()
        while not self.stop:
            s = smi()
            s["t"] = round(time.perf_counter() - t0, 2)
            self.rows.append(s)
            time.sleep(self.period)

    def __enter__(self):
        self.t.start()
        return self

    def __exit__(self, *a):
        self.stop = True
        self.t.join(timeout=5)


def num(v):
    m = re.search(r"[-+]?\d+\.?\d*", str(v))
    return float(m.group(0)) if m else None


def clock_summary(rows):
    out = {}
    for key in ("sclk", "mclk", "fclk", "Power"):
        vals = [num(v) for r in rows for k, v in r.items() if key in k and num(v) is not None]
        if vals:
            out[key] = {"min": min(vals), "max": max(vals), "mean": sum(vals) / len(vals), "n": len(vals)}
    return out


def run(tag, cmd, prof=None, timeout=600):
    d = os.path.join(OUT, tag)
    subprocess.run(["rm", "-rf", d])
    full = (["rocprofv3"] + prof + ["--output-format", "csv", "-d", d, "-o", "p", "--"] if prof else []) + cmd
    before = smi()
    with Sampler() as sm:
        t0 = time.perf_counter()
        try:
            res = subprocess.run(full, cwd="/tmp", env=ENV, capture_output=True, text=True, timeout=timeout)
            rc, so, se = res.returncode, res.stdout, res.stderr
        except subprocess.TimeoutExpired:
            rc, so, se = -9, "", "TIMEOUT"
        wall = time.perf_counter() - t0
    after = smi()
    open(os.path.join(OUT, tag + ".log"), "w").write(so[-20000:] + "\n---- stderr ----\n" + se[-4000:])
    line = next((l for l in reversed(so.splitlines()) if l.startswith("{")), None)
    rec = {"rc": rc, "wall_s": wall, "smi_before": before, "smi_after": after, "smi_during": clock_summary(sm.rows), "smi_samples": len(sm.rows),
           "line": json.loads(line) if line else None}
    print(tag, "rc", rc, "wall %.1f s" % wall, flush=True)
    return rec


def find(tag, pattern):
    return next(iter(glob.glob(os.path.join(OUT, tag, "**", pattern), recursive=True)), None)


def by_grid(tag, frag):
    f = find(tag, "*kernel_trace.csv")
    agg = collections.defaultdict(list)
    if f:
        for r in csv.DictReader(open(f)):
            name = r.get("Kernel_Name", "")
            if frag in name:
                agg[(re.sub(r"\(.*", "", name)[:90], int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0))].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    rows = [{"Name": n, "Grid_Size_X": g, "Calls": len(v), "TotalNs": sum(v), "AverageNs": sum(v) / len(v), "MinNs": min(v), "MaxNs": max(v),
             "AverageNs_last_three_quarters": sum(v[len(v) // 4:]) / max(1, len(v[len(v) // 4:]))}
            for (n, g), v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))]
    if rows:
        with open(os.path.join(OUT, tag + "_kernel_stats_by_grid.csv"), "w", newline="") as g:
            wr = csv.DictWriter(g, fieldnames=list(rows[0].keys()))
            wr.writeheader()
            wr.writerows(rows)
    return rows


def counter_mean(tag, frag, name):
    f = find(tag, "*counter_collection.csv")
    vals = []
    if f:
        for row in csv.DictReader(open(f)):
            if frag in row["Kernel_Name"] and row["Counter_Name"] == name:
                vals.append(float(row["Counter_Value"]))
    vals = vals[len(vals) // 4:]
    return (sum(vals) / len(vals), len(vals)) if vals else (None, 0)


def brief(line):
    if not line:
        return None
    roof = line["roofline"]
    return {"launch_us_hip_events": roof["launch_us_hip_events"], "frac": roof["frac"], "achieved_gbs": roof["achieved"],
            "bytes_per_launch": roof["bytes_per_launch"], "launches_timed": roof["launches_timed"], "timed_region_ms": line.get("timed_region_ms"),
            "segments": line.get("segments"), "clock_ghz_after": line.get("clock_ghz_after_timed_region"),
            "parity_max": (line.get("parity") or {}).get("max")}


def copy_probe():
    """(--copy-probe) one JSON line: device-to-device copy rate, HIP events"""
    import torch
    sys.path.insert(0, R)
    import bench
    dev = torch.device("cuda", 0)
    torch.zeros(1, device=dev)
    r = bench.copy_probe(dev, mib=1024, reps=200)
    print(json.dumps(r))


def main():
    S = {"what": __doc__.split("\n\n")[0], "legs": {}}
    S["legs"]["a_plain"] = run("a_plain", BENCH + ["--min-time", "2.0"])
    S["legs"]["a_copy_plain"] = run("a_copy_plain", COPY)
    S["legs"]["b_traced"] = run("b_traced", BENCH + ["--min-time", "2.0"], ["--kernel-trace", "--stats"])
    S["by_grid_traced"] = by_grid("b_traced", "gpd_rollout")
    S["legs"]["b_copy_traced"] = run("b_copy_traced", COPY, ["--kernel-trace"])
    S["copy_by_grid_traced"] = by_grid("b_copy_traced", "")[:4]
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        tag = "c_pmc_" + ctr
        S["legs"][tag] = run(tag, BENCH + ["--min-time", "0.05", "--no-parity"], ["--kernel-trace", "--pmc", ctr])
        m, n = counter_mean(tag, "gpd_rollout", ctr)
        S["legs"][tag]["counter_mean_per_dispatch"] = m
        S["legs"][tag]["dispatches"] = n
        rows = by_grid(tag, "gpd_rollout")
        S["legs"][tag]["kernel_avg_ns_in_pmc_pass"] = rows[0]["AverageNs"] if rows else None
    S["legs"]["d_plain"] = run("d_plain", BENCH + ["--min-time", "2.0"])
    for k, v in S["legs"].items():
        v["brief"] = brief(v["line"]) if v.get("line") and "roofline" in (v["line"] or {}) else v.get("line")
    json.dump(S, open(os.path.join(OUT, "summary.json"), "w"), indent=1)
    for d in glob.glob(os.path.join(OUT, "*")):
        if os.path.isdir(d):
            shutil.rmtree(d, ignore_errors=True)
    for k, v in S["legs"].items():
        b = v.get("brief") or {}
        print(k, {q: b.get(q) for q in ("launch_us_hip_events", "frac", "clock_ghz_after", "gbs")}, v["smi_during"].get("sclk"), v["smi_during"].get("mclk"), v["smi_during"].get("Power"))
    for r in S["by_grid_traced"][:3]:
        print("traced", r["Name"][:60], r["Grid_Size_X"], r["Calls"], "avg us %.1f" % (r["AverageNs"] / 1e3))


if __name__ == "__main__":
    if "--copy-probe" in sys.argv:
        copy_probe()
    else:
        main()
