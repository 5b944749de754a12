#!/usr/bin/env python3
"""A/B of `gpd_step_kernel` variants on ONE box (VERDICT r04 "next" #2): bench.py's one-launch-per-step leg (hipGraph of 64 gpd_step
launches) per library, interleaved over several rounds so that box drift shows as spread, not as a difference.

  v0  the kernel with its arguments fetched by scalar loads (no kernarg preload)
  v1  kernarg preload: the first 14 argument dwords (what the load section needs) arrive in SGPRs with the wave
  v2  v1 + the kinematic state as three float4 planes + one float row (`-DGPD_EXP_KIN4`; same 13 x ld floats): 4 loads + 4 stores
      per lane instead of 13 + 13

usage (GPU box): python scratch/exp_r05/ab_step.py [rounds]   -> gpurun_out/ab_step_r05.json + a table on stdout"""
import json
import os
import subprocess
import sys

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
LIBS = {"v1": ("gym-pybullet-drones_amd/csrc/libgpd.so", {}),
        "v2": ("scratch/exp/libgpd_v2.so", {"GPD_EXP_KIN4": "1"}),
        "v3": ("scratch/exp/libgpd_v3.so", {"GPD_EXP_KIN4": "1"}),       # v2 + the state stored non-temporally at every size
        "v4": ("scratch/exp/libgpd_v4.so", {})}                         # v1 + the state stored non-temporally at every size
WORK = [("hover65536_240hz", ["--min-time", "0.5"]), ("hover65536_pid_240hz", ["--min-time", "0.3"]), ("hover65536_30hz", ["--min-time", "0.3"]),
        ("hover4096_240hz", ["--min-time", "0.3"]), ("hover4m_240hz", ["--min-time", "0.3"])]
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
res = {}
for rnd in range(rounds):
    for wl, extra in WORK:
        if rnd and wl not in ("hover65536_240hz", "hover4m_240hz"):
            continue
        for v, (lib, env) in LIBS.items():
            if not os.path.exists(os.path.join(R, lib)):
                continue
            e = dict(os.environ, GPD_LIB=os.path.join(R, lib), **env)
            cmd = [sys.executable, os.path.join(R, "bench.py"), "--workload", wl, "--mode", "graph", "--steps", "64", "--warmup", "64",
                   "--no-cpu-baseline", "--no-hbm-leg", "--no-parity"] + extra
            p = subprocess.run(cmd, env=e, capture_output=True, text=True, timeout=300)
            line = next((l for l in reversed(p.stdout.splitlines()) if l.startswith("{")), None)
            if not line:
                print(wl, v, "FAILED", p.stderr[-400:], flush=True)
                continue
            j = json.loads(line)
            us = j["ms_per_step"] * 1e3
            res.setdefault(wl, {}).setdefault(v, []).append(us)
            print(f"round {rnd} {wl:24s} {v}: {us:.3f} us/step  frac {j['roofline']['frac']:.3f}", flush=True)
os.makedirs(os.path.join(R, "gpurun_out"), exist_ok=True)
json.dump(res, open(os.path.join(R, "gpurun_out", "ab_step_r05.json"), "w"), indent=1)
print("\nus per step (min .. max over rounds)")
for wl, d in res.items():
    print(f"{wl:24s} " + "   ".join(f"{v}: {min(x):.3f}..{max(x):.3f}" for v, x in d.items()))
