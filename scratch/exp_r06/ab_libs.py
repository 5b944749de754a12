#!/usr/bin/env python3
"""A/B of variant libraries through bench.py (round 6: the product vs scratch/exp_r06/pair_z.patch),
interleaved rounds on one box.  usage (GPU box): python scratch/exp_r05/ab_rollout.py [rounds] -> gpurun_out/ab_libs_r06.log
(GPD_AB_LIBS="a=path,b=path" compares other variant libraries; GPD_AB_QUICK=1: the two headline shapes only)"""
import json
import os
import subprocess
import sys

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
LIBS = {"product": "gym_pybullet_drones_amd/csrc/libgpd.so", "pair_z": "scratch/exp_r06/libgpd_pairz.so"}
if os.environ.get("GPD_AB_LIBS"):          # "label=path,label=path" (paths relative to the repo): any two (or more) variant libraries
    LIBS = dict(kv.split("=", 1) for kv in os.environ["GPD_AB_LIBS"].split(","))
WORK = [("hover65536_240hz", ["--steps", "20", "--warmup", "5"]), ("hover65536_240hz", ["--steps", "64", "--warmup", "64"]),
        ("hover65536_pid_240hz", ["--steps", "20", "--warmup", "5"]), ("hover65536_30hz", ["--steps", "20", "--warmup", "5"]),
        ("stack8x8192_ext_240hz", ["--steps", "20", "--warmup", "5"]), ("hover4096_240hz", ["--steps", "20", "--warmup", "5"])]
if os.environ.get("GPD_AB_QUICK"):         # the two headline shapes only
    WORK = WORK[:2]
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
res = {}
for rnd in range(rounds):
    for wl, extra in WORK:
        if rnd and wl != "hover65536_240hz" and not os.environ.get("GPD_AB_MODE"):
            continue
        for v, lib in LIBS.items():
            e = dict(os.environ, GPD_LIB=os.path.join(R, lib))
            cmd = [sys.executable, os.path.join(R, "bench.py"), "--workload", wl, "--no-cpu-baseline", "--no-hbm-leg", "--no-parity", "--no-second-leg", "--no-dropin-leg",
                   "--min-time", "0.5"] + extra + (["--mode", os.environ["GPD_AB_MODE"]] if os.environ.get("GPD_AB_MODE") else [])   # (graph: the single-step kernel)
            p = subprocess.run(cmd, env=e, capture_output=True, text=True, timeout=300)
            line = next((l for l in reversed(p.stdout.splitlines()) if l.startswith("{")), None)
            if not line:
                print(wl, v, "FAILED", p.stderr[-300:], flush=True)
                continue
            j = json.loads(line)
            key = f"{wl} K={extra[1]}"
            res.setdefault(key, {}).setdefault(v, []).append(j["roofline"]["launch_us_hip_events"])
            print(f"round {rnd} {key:32s} {v:8s}: {j['roofline']['launch_us_hip_events']:.3f} us per launch, {j['ms_per_step'] * 1e3:.4f} us per step, frac {j['roofline']['frac']:.3f}", flush=True)
print("\nus per launch (min .. max over rounds)")
for k, d in res.items():
    print(f"{k:34s} " + "   ".join(f"{v}: {min(x):.3f}..{max(x):.3f}" for v, x in d.items()))
