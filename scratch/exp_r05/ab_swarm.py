#!/usr/bin/env python3
"""A/B of the pair-balanced group placement (GPD_SWARM_BALANCE=0 / 1) on ONE box: bench.py's one-world workloads, interleaved rounds,
then `rocprofv3 --kernel-trace --stats` of the 65 536-drone workload both ways (per-kernel durations: dwg_force_kernel<2> is the replay).
usage (GPU box): python scratch/exp_r05/ab_swarm.py [rounds]  -> gpurun_out/ab_swarm_r05.json / .log"""
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
OUT = os.path.join(R, "gpurun_out")
os.makedirs(OUT, exist_ok=True)
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
WORK = [("swarm65536_ext_240hz", ["--steps", "240", "--warmup", "24"]), ("swarm1m_ext_240hz", ["--steps", "64", "--warmup", "16"])]
res = {}


def bench(wl, extra, bal, prof=None):
    env = dict(os.environ, GPD_SWARM_BALANCE=str(bal), TMPDIR="/tmp")
    cmd = [sys.executable, os.path.join(R, "bench.py"), "--workload", wl, "--no-cpu-baseline", "--no-parity"] + extra
    if prof:
        cmd = ["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", prof, "-o", "p", "--"] + cmd
    p = subprocess.run(cmd, env=env, cwd="/tmp", capture_output=True, text=True, timeout=600)
    line = next((l for l in reversed(p.stdout.splitlines()) if l.startswith("{")), None)
    if not line:
        print(wl, bal, "FAILED", p.stderr[-600:], flush=True)
        return None
    return json.loads(line)


for rnd in range(rounds):
    for wl, extra in WORK:
        if rnd and wl == "swarm1m_ext_240hz" and rnd > 1:
            continue
        for bal in (0, 1):
            j = bench(wl, extra, bal)
            if j:
                us = j["ms_per_step"] * 1e3
                res.setdefault(wl, {}).setdefault(f"balance={bal}", []).append(us)
                print(f"round {rnd} {wl:22s} balance={bal}: {us:.3f} us per sub-step", flush=True)
stats = {}
for bal in (0, 1):
    d = os.path.join(OUT, f"prof_swarm_bal{bal}")
    subprocess.run(["rm", "-rf", d])
    j = bench("swarm65536_ext_240hz", ["--steps", "240", "--warmup", "24"], bal, prof=d)
    f = next(iter(glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)), None)
    rows = [r for r in csv.DictReader(open(f)) if "dwg_" in r["Name"] or "gpd_swarm" in r["Name"]] if f else []
    stats[f"balance={bal}"] = {"us_per_substep_under_rocprof": j["ms_per_step"] * 1e3 if j else None,
                               "kernels": [{"name": r["Name"][:80], "calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3} for r in rows]}
    if rows:
        with open(os.path.join(OUT, f"ab_swarm_r05_bal{bal}_kernel_stats.csv"), "w", newline="") as g:
            wr = csv.DictWriter(g, fieldnames=list(rows[0].keys()))
            wr.writeheader()
            wr.writerows(rows)
    shutil.rmtree(d, ignore_errors=True)
    for k in stats[f"balance={bal}"]["kernels"]:
        print(f"balance={bal} {k['name'][:60]:60s} calls {k['calls']:6d} avg {k['avg_us']:.2f} us", flush=True)
json.dump({"us_per_substep": res, "rocprof": stats}, open(os.path.join(OUT, "ab_swarm_r05.json"), "w"), indent=1)
print("\nus per sub-step (min .. max over rounds)")
for wl, d in res.items():
    print(f"{wl:22s} " + "   ".join(f"{v}: {min(x):.3f}..{max(x):.3f}" for v, x in d.items()))
