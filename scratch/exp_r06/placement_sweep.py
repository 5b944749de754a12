#!/usr/bin/env python3
"""Round 6, second placement experiment.  The first one (placement_cause.py, profiles/r06_hbm_placement_cause.md) showed: same request
counts, balanced channels, identical TLB behaviour, but 40x the L2 tag-pipeline stall cycles in a slow placement, and that WHICH action
block a given observation block is paired with decides the level.  Here: ONE allocation holds every block of the launch, and only the
OFFSETS of the blocks inside it change between measurements -- the physical memory stays where it is.

    python scratch/exp_r06/placement_sweep.py [--slabs 2] [--tag NAME]
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
ap.add_argument("--slabs", type=int, default=2)
ap.add_argument("--tag", default="sweep")
ap.add_argument("--launches", type=int, default=6)
ap.add_argument("--E", type=int, default=4194304)
ap.add_argument("--K", type=int, default=64)
args = ap.parse_args()
dev = torch.device("cuda:0")
E, K = args.E, args.K
rng = np.random.default_rng(0)
MiB, GiB = 1 << 20, 1 << 30


def timed(fn, reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / reps


xyz = np.array([0, 0, 0.1125]) + rng.uniform(-0.5, 0.5, size=(E, 1, 3)) * np.array([1, 1, 0])
env = VectorAviary(E, 1, initial_xyzs=xyz, initial_rpys=rng.uniform(-0.1, 0.1, size=(E, 1, 3)), physics=0, pyb_freq=240, ctrl_freq=240,
                   act=ActionType.RPM, task="hover", auto_reset=True, track_rpm=False, device=dev)
core = env.core
sizes = {"obs": K * E * 48, "act": K * E * 16, "rew": K * E * 4, "term": K * E, "trunc": K * E}
total = sum(sizes.values()) + 8 * GiB
# layouts: name -> start offset of every block inside the slab (bytes)
base = {"obs": 0, "act": 12 * GiB, "rew": 16 * GiB, "term": 17 * GiB, "trunc": 17 * GiB + 256 * MiB}
layouts = {"packed": base}
for name, d in (("act+4K", 4096), ("act+64K", 64 << 10), ("act+1M", MiB), ("act+2M", 2 * MiB), ("act+4M", 4 * MiB), ("act+32M", 32 * MiB), ("act+256M", 256 * MiB), ("act+1G", GiB)):
    layouts[name] = dict(base, act=base["act"] + d, rew=base["rew"] + 2 * GiB, term=base["term"] + 2 * GiB, trunc=base["trunc"] + 2 * GiB)
layouts["act_first"] = {"act": 0, "obs": 4 * GiB + 4 * MiB, "rew": 17 * GiB, "term": 18 * GiB + 2 * MiB, "trunc": 18 * GiB + 300 * MiB}
layouts["small_first"] = {"rew": 0, "term": GiB + 2 * MiB, "trunc": GiB + 300 * MiB, "act": 2 * GiB, "obs": 6 * GiB + 4 * MiB}
layouts["torch_like"] = dict(base, act=16 * GiB + 4 * MiB, rew=20 * GiB + 6 * MiB, term=21 * GiB + 8 * MiB, trunc=21 * GiB + 300 * MiB)
rows = []
for s in range(args.slabs):
    slab = torch.empty(total, dtype=torch.uint8, device=dev)
    for name, lay in layouts.items():
        cut = lambda k, dt, shape: slab[lay[k]:lay[k] + sizes[k]].view(dt).view(shape)      # noqa: E731
        obs, acts = cut("obs", torch.float32, (K, E, 12)), cut("act", torch.float32, (K, E, 1, 4))
        rew, term, trunc = cut("rew", torch.float32, (K, E)), cut("term", torch.bool, (K, E)), cut("trunc", torch.bool, (K, E))
        acts.uniform_(-1, 1)
        core.__dict__["_rollout_cache"] = {K: (obs, rew, term, trunc, None)}
        core.reset()
        core.rollout(acts, update_latest=False)
        core.rollout(acts, update_latest=False)
        sec = timed(lambda: core.rollout(acts, update_latest=False), args.launches)
        row = {"slab": s, "slab_ptr": hex(slab.data_ptr()), "layout": name, "us_per_launch": sec * 1e6, "frac": core.bytes_per_rollout(K) / sec / 8e12}
        rows.append(row)
        print(json.dumps(row), flush=True)
    del slab, obs, acts, rew, term, trunc
    core.__dict__["_rollout_cache"] = {}
    gc.collect()
    torch.cuda.empty_cache()
    keep = torch.empty(int(rng.integers(500, 3000)) << 20, dtype=torch.uint8, device=dev)        # shift the next slab
json.dump(rows, open(os.path.join(R, "gpurun_out", f"placement_{args.tag}.json"), "w"), indent=1)
