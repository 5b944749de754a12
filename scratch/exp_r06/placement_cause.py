#!/usr/bin/env python3
"""Round 6, VERDICT r05 #2: WHY does the hover4m rollout leg (gpd_rollout1_kernel, 4 194 304 drones, 64 env steps per launch, 19.26 GB per
launch) run at 0.62 / 0.68 / 0.72 / 0.75 of 8 TB/s depending on the allocation?  One process, TRIALS fresh allocations of everything the
launch touches; per trial: the launch's own rate (HIP events), a write-only stream over ITS observation block, a read stream over ITS
action block, the device addresses.  Run plain, and under `rocprofv3 --pmc ...` (every dispatch's counters + timestamps land in
rocprofv3's csv; this script prints which dispatch numbers belong to which trial).

    python scratch/exp_r06/placement_cause.py [--trials 8] [--mode torch|slab|retry] [--tag NAME]

mode torch : every buffer its own torch allocation (what bench.py's hbm leg does)
mode slab  : observation / action / reward / flag blocks carved out of ONE allocation
mode retry : like torch, but an observation block whose own write stream is slow is kept aside and another one is allocated
"""
import argparse
import gc
import json
import os
import sys

import numpy as np
import torch

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
from gym_pybullet_drones_amd.envs import VectorAviary  # noqa: E402
from gym_pybullet_drones_amd.utils.enums import ActionType  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--trials", type=int, default=8)
ap.add_argument("--mode", default="torch")
ap.add_argument("--tag", default="plain")
ap.add_argument("--launches", type=int, default=12)
ap.add_argument("--E", type=int, default=4194304)
ap.add_argument("--K", type=int, default=64)
ap.add_argument("--retry-below", type=float, default=0.0, help="mode retry: re-allocate the obs block while its write stream is below this many GB/s")
args = ap.parse_args()
dev = torch.device("cuda:0")
E, K = args.E, args.K
rng = np.random.default_rng(0)


def timed(fn, reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / reps


read_dst = torch.empty(K * E * 4, dtype=torch.float32, device=dev)      # the fixed end of the read probe (same block in every trial)
held, rows, n_disp = [], [], 0
for trial in range(args.trials):
    if trial and trial % 2 == 0:                                   # shift what the driver hands out next
        held.append(torch.empty(int(rng.integers(200, 3000)) << 20, dtype=torch.uint8, device=dev))
    xyz = np.array([0, 0, 0.1125]) + rng.uniform(-0.5, 0.5, size=(E, 1, 3)) * np.array([1, 1, 0])
    env = VectorAviary(E, 1, initial_xyzs=xyz, initial_rpys=rng.uniform(-0.1, 0.1, size=(E, 1, 3)), physics=0, pyb_freq=240, ctrl_freq=240,
                       act=ActionType.RPM, task="hover", auto_reset=True, track_rpm=False, device=dev)
    core = env.core
    aside = []
    mode = args.mode if args.mode != "alt" else ("slab" if trial % 2 else "torch")
    if mode == "slab":
        sizes = [K * E * 12 * 4, K * E * 4 * 4, K * E * 4, K * E, K * E]
        offs = np.cumsum([0] + [(s + 4095) // 4096 * 4096 for s in sizes])
        slab = torch.empty(int(offs[-1]), dtype=torch.uint8, device=dev)
        cut = lambda i, dt, shape: slab[int(offs[i]):int(offs[i]) + sizes[i]].view(dt).view(shape)      # noqa: E731
        obs, acts = cut(0, torch.float32, (K, E, 12)), cut(1, torch.float32, (K, E, 1, 4))
        rew, term, trunc = cut(2, torch.float32, (K, E)), cut(3, torch.bool, (K, E)), cut(4, torch.bool, (K, E))
        acts.uniform_(-1, 1)
        core.__dict__.setdefault("_rollout_cache", {})[K] = (obs, rew, term, trunc, None)
    else:
        acts = torch.rand((K, E, 1, 4), device=dev) * 2 - 1
        obs = None
        if mode == "retry":
            while True:
                obs = torch.empty((K, E, 12), dtype=torch.float32, device=dev)
                obs.fill_(0.0)
                w = obs.numel() * 4 / timed(lambda: obs.fill_(0.0), 3) / 1e9
                if w >= args.retry_below or len(aside) >= 6:
                    break
                aside.append(obs)
            core.__dict__.setdefault("_rollout_cache", {})[K] = (obs, torch.zeros((K, E), device=dev), torch.zeros((K, E), dtype=torch.bool, device=dev),
                                                                torch.zeros((K, E), dtype=torch.bool, device=dev), None)
    first = n_disp
    out = core.rollout(acts, update_latest=False)
    for _ in range(3):
        core.rollout(acts, update_latest=False)
    sec = timed(lambda: core.rollout(acts, update_latest=False), args.launches)
    n_disp += 4 + args.launches
    obs = out[0]
    bytes_launch = core.bytes_per_rollout(K)
    obs.fill_(0.0)
    w_gbs = obs.numel() * 4 / timed(lambda: obs.fill_(0.0), 3) / 1e9
    flat = acts.view(-1)
    read_dst.copy_(flat)
    r_gbs = 2 * flat.numel() * 4 / timed(lambda: read_dst.copy_(flat), 3) / 1e9
    row = {"trial": trial, "us_per_launch": sec * 1e6, "frac": bytes_launch / sec / 8e12, "obs_fill_gbs": w_gbs, "actions_copy_gbs": r_gbs,
           "rollout_dispatches": [first, n_disp - 1], "set_aside": len(aside), "mode": mode,
           "ptr": {"obs": hex(obs.data_ptr()), "actions": hex(acts.data_ptr()), "reward": hex(out[1].data_ptr()), "state": hex(core.kin_store.data_ptr())}}
    rows.append(row)
    print(json.dumps(row), flush=True)
    del env, core, acts, obs, out, aside
    mode = args.mode if args.mode != "alt" else ("slab" if trial % 2 else "torch")
    if mode == "slab":
        del slab, rew, term, trunc
    gc.collect()
    torch.cuda.empty_cache()
fr = [r["frac"] for r in rows]
print(json.dumps({"tag": args.tag, "mode": args.mode, "fracs": fr, "min": min(fr), "max": max(fr),
                  "corr_frac_vs_obs_fill": float(np.corrcoef(fr, [r["obs_fill_gbs"] for r in rows])[0, 1]) if len(fr) > 2 else None,
                  "corr_frac_vs_actions_copy": float(np.corrcoef(fr, [r["actions_copy_gbs"] for r in rows])[0, 1]) if len(fr) > 2 else None}), flush=True)
os.makedirs(os.path.join(R, "gpurun_out"), exist_ok=True)
json.dump(rows, open(os.path.join(R, "gpurun_out", f"placement_{args.tag}.json"), "w"), indent=1)
