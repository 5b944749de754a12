"""Round 6, final tree: the driver's own command plain and under `rocprofv3 --kernel-trace --stats` on ONE lease, and the figures
the bench line quotes from the trace (profiles/r06_hbm_reconcile.json, profiles/r06_trace_driver_cmd_*).

    python scratch/exp_r06/trace_driver_cmd.py            # on a GPU box; writes gpurun_out/r06_final/

For every (kernel, grid size) the tracer's own durations are averaged (a) over all dispatches and (b) over the TIMED REGION of
that grid size = the longest run of consecutive dispatches of that kernel and grid that no other kernel interrupts (gaps do not
end a run: under the tracer the host falls behind the 20 us launches now and then, which the HIP events see and the kernel
durations do not); (b) is what the traced process's HIP events bracket."""
import collections
import csv
import glob
import json
import os
import subprocess
import sys

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
OUT = os.path.join(R, "gpurun_out", "r06_final")
os.makedirs(OUT, exist_ok=True)
ENV = dict(os.environ, TMPDIR="/tmp")
CMD = ["--steps", "20", "--warmup", "5"]


def bench(tag, prefix=(), extra=()):
    res = subprocess.run(list(prefix) + [sys.executable, os.path.join(R, "bench.py")] + CMD + list(extra), cwd="/tmp", env=ENV,
                         capture_output=True, text=True, timeout=900)
    line = next((l for l in reversed(res.stdout.splitlines()) if l.startswith("{")), None)
    open(os.path.join(OUT, tag + ".err"), "w").write(res.stderr[-8000:])
    if line:
        open(os.path.join(OUT, tag + ".json"), "w").write(line + "\n")
    print(tag, "rc", res.returncode, flush=True)
    return json.loads(line) if line else None


def short(name):
    for k in ("rollout1", "rollout_kernel", "step_kernel", "reset", "state20", "hist"):
        if k in name:
            return k
    return name.split("(")[0][-40:]


def main():
    plain = bench("bench_driver_cmd")
    d = os.path.join(OUT, "trace")
    subprocess.run(["rm", "-rf", d])
    traced = bench("bench_driver_cmd_under_rocprofv3",
                   prefix=["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", d, "-o", "p", "--"],
                   extra=["--no-cpu-baseline"])
    f = next(iter(glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)), None)
    if f:
        rows = [r for r in csv.DictReader(open(f)) if "gpd_" in r["Name"]]
        with open(os.path.join(OUT, "trace_driver_cmd_kernel_stats.csv"), "w", newline="") as g:
            wr = csv.DictWriter(g, fieldnames=list(rows[0].keys()))
            wr.writeheader()
            wr.writerows(rows)
    f = next(iter(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)), None)
    disp = []
    for r in csv.DictReader(open(f)):
        disp.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Kernel_Name", ""), int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0)))
    disp.sort()
    agg = collections.defaultdict(list)
    runs = collections.defaultdict(list)             # (kernel, grid) -> list of runs (each a list of durations)
    prev_key, prev_end = None, None
    for s, e, name, grid in disp:
        if "gpd_" not in name:
            prev_key = None
            continue
        key = (short(name), grid)
        agg[key].append(e - s)
        if key != prev_key:
            runs[key].append([])
        runs[key][-1].append(e - s)
        prev_key, prev_end = key, e
    by_grid = []
    for (k, g), v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        longest = max(runs[(k, g)], key=len)
        by_grid.append({"kernel": k, "grid_threads": g, "calls": len(v), "avg_us": sum(v) / len(v) / 1e3, "min_us": min(v) / 1e3, "max_us": max(v) / 1e3,
                        "timed_region_launches": len(longest), "timed_region_avg_us": sum(longest) / len(longest) / 1e3})
    json.dump(by_grid, open(os.path.join(OUT, "trace_driver_cmd_by_grid.json"), "w"), indent=1)
    rec = {"what": "round 6, final tree: `rocprofv3 --kernel-trace --stats -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline` on one MI355X lease "
                   "(scratch/exp_r06/trace_driver_cmd.py); the tracer's own durations of the launches of each timed region (the longest run of launches of "
                   "that kernel and grid size that nothing interrupts) against the HIP events the traced process printed for the same region "
                   "(profiles/r06_bench_driver_cmd_under_rocprofv3.json); the untraced run of the same command in the same lease: profiles/r06_bench_driver_cmd.json",
           "keys": {}}
    if traced:
        def region(grid):
            return next((b for b in by_grid if b["kernel"] == "rollout1" and b["grid_threads"] == grid), None)
        h, hb = region(4194304), traced.get("hbm_saturating", {})
        if h and "bytes_per_launch" in hb:
            rec["keys"]["hover4m_240hz:rollout64"] = {
                "rocprof_kernel_avg_us": h["timed_region_avg_us"], "rocprof_frac": hb["bytes_per_launch"] / (h["timed_region_avg_us"] * 1e-6) / 8e12,
                "trace_launches": h["timed_region_launches"], "hip_events_us_same_process": hb["launch_us_hip_events"], "hip_events_frac_same_process": hb["frac"],
                "note": "blocks placed by RolloutArena.search in that process; the %d launches of the >= 2 s timed region: tracer %.1f us, the process's own HIP events %.1f us"
                        % (h["timed_region_launches"], h["timed_region_avg_us"], hb["launch_us_hip_events"])}
        s, rf = region(65536), traced["roofline"]
        if s:
            rec["keys"]["hover65536_240hz:rollout20"] = {
                "rocprof_kernel_avg_us": s["timed_region_avg_us"], "rocprof_frac": rf["bytes_per_launch"] / (s["timed_region_avg_us"] * 1e-6) / 8e12,
                "trace_launches": s["timed_region_launches"], "hip_events_us_same_process": rf["launch_us_hip_events"],
                "note": "the %d launches of the headline's timed region in the traced run of the driver's own command: the tracer's durations average %.2f us, that "
                        "process's HIP events %.2f us per launch (the events include the gaps between launches)"
                        % (s["timed_region_launches"], s["timed_region_avg_us"], rf["launch_us_hip_events"])}
    json.dump(rec, open(os.path.join(OUT, "hbm_reconcile.json"), "w"), indent=1)
    subprocess.run(["rm", "-rf", d])
    for b in by_grid[:6]:
        print(b)
    for tag, j in (("plain", plain), ("traced", traced)):
        if j:
            print(tag, j["value"], j["roofline"]["frac"], j["roofline"]["launch_us_hip_events"], j.get("hbm_saturating", {}).get("frac"),
                  j.get("hbm_saturating", {}).get("frac_range_over_allocations"))


if __name__ == "__main__":
    main()
