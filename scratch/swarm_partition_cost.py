#!/usr/bin/env python3
"""Round 3: how well does the force work of ONE world split across ranks?  W ranks of one process on ONE GPU (LocalSwarmGroup),
their launches one after the other on one stream: the time of a step summed over the ranks, against the single-rank step.
A perfect split keeps the sum constant (what every rank repeats -- the binning of all rows -- grows with W)."""
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402
from gym_pybullet_drones_amd.envs import LocalSwarmGroup, SwarmAviary  # noqa: E402
from gym_pybullet_drones_amd.utils.enums import Physics  # noqa: E402

dev = torch.device("cuda:0")
for D in (65536, 1048576):
    w = dict(bench.WORKLOADS["swarm65536_ext_240hz"], D=D)
    one = bench.make_env(w, dev, seed=1000)
    xyz, rpy = one.INIT_XYZS.copy(), one.INIT_RPYS.copy()
    rpm = torch.full((D, 4), float(one.HOVER_RPM), device=dev)
    kw = dict(initial_xyzs=xyz, initial_rpys=rpy, physics=Physics.PYB_GND_DRAG_DW, pyb_freq=240, ctrl_freq=240, device=dev)

    def timed(env, steps=48):
        env.reset()
        for _ in range(8):
            env.step(rpm)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(steps):
            env.step(rpm)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / steps

    base = timed(one)
    print(f"N {D}: single rank {base:.1f} us per step (eager launches, host-bound below ~80 us)", flush=True)
    del one
    for part in ("spatial", "index"):
        for W in (2, 4, 8):
            grp = LocalSwarmGroup(D, W, partition=part, **kw)
            t = timed(grp)
            print(f"   {part:7s} W = {W}: {t:8.1f} us per step summed over the ranks = {t / base:.2f} x the single rank, {t / W:.1f} us per rank", flush=True)
            del grp
            torch.cuda.empty_cache()
