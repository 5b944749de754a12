#!/usr/bin/env python3
"""Does the rate of the HBM-served rollout leg (4 194 304 drones, 64 steps per launch) depend on the STEP STRIDES of its output / input
blocks?  With contiguous [K][N][12] rows the 64 step planes a launch writes concurrently lie exactly 3 * 2^26 bytes apart (and the
reward / flag / action planes 2^24 / 2^22 / 2^26 bytes): every plane presents the same low address bits to the memory channels at the
same time.  gpd_rollout takes the strides as arguments, so padded blocks need no kernel change.

usage (GPU box): python scratch/exp_r05/stride_probe.py [rounds] > gpurun_out/stride_probe.txt"""
import ctypes
import gc
import os
import sys

import torch

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
from gym_pybullet_drones_amd import _native  # noqa: E402
from gym_pybullet_drones_amd.engine import _ptr  # noqa: E402
from gym_pybullet_drones_amd.envs import VectorAviary  # noqa: E402
from gym_pybullet_drones_amd.utils.enums import ActionType  # noqa: E402

E, K = 4194304, 64
PADS = [0, 64, 1024, 16384 + 64, 262144 + 1024 + 64]          # floats (x 4 bytes) added to every step stride
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
dev = torch.device("cuda:0")


def one(pad):
    import numpy as np
    rng = np.random.default_rng(0)          # (bench.py's hover4m_240hz scene)
    xyz = np.array([0, 0, 0.1125]) + rng.uniform(-0.5, 0.5, size=(E, 1, 3)) * np.array([1, 1, 0])
    env = VectorAviary(E, 1, initial_xyzs=xyz, initial_rpys=rng.uniform(-0.1, 0.1, size=(E, 1, 3)), physics=0, pyb_freq=240, ctrl_freq=240,
                       act=ActionType("rpm"), task="hover", auto_reset=True, track_rpm=False, device=dev)
    env.reset()
    core = env.core
    N, A = core.N, core.A
    a_stride, o_stride, e_stride = N * A + pad, N * 12 + pad, E + pad
    g = torch.Generator(device=dev).manual_seed(0)
    act = torch.empty((K, a_stride), dtype=torch.float32, device=dev).uniform_(-1, 1, generator=g)
    obs = torch.empty((K, o_stride), dtype=torch.float32, device=dev)
    rew = torch.empty((K, e_stride), dtype=torch.float32, device=dev)
    term = torch.empty((K, e_stride), dtype=torch.uint8, device=dev)
    trunc = torch.empty((K, e_stride), dtype=torch.uint8, device=dev)

    def launch():
        rc = core.lib.gpd_rollout(ctypes.byref(core._params), ctypes.byref(core._state), ctypes.byref(core._cfg), K, _ptr(act), a_stride,
                                  _ptr(core.target), _ptr(core.init_pose), _ptr(obs), o_stride, _ptr(rew), _ptr(term), _ptr(trunc), e_stride,
                                  None, core._stream())
        _native.check(rc, "gpd_rollout")

    with torch.cuda.device(dev):
        for _ in range(3):
            launch()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        ev0.record()
        reps = 40
        for _ in range(reps):
            launch()
        ev1.record()
        torch.cuda.synchronize()
    us = ev0.elapsed_time(ev1) * 1e3 / reps
    frac = core.bytes_per_rollout(K) / (us * 1e-6) / 8e12
    del env, core, act, obs, rew, term, trunc
    gc.collect()
    torch.cuda.empty_cache()
    return us, frac


res = {p: [] for p in PADS}
for r in range(rounds):
    for p in PADS:
        us, frac = one(p)
        res[p].append(frac)
        print(f"round {r} pad {p:7d} floats: {us:8.1f} us per launch, frac {frac:.3f}", flush=True)
print("\nfrac of 8 TB/s, min .. max over rounds (fresh allocations every time)")
for p, v in res.items():
    print(f"pad {p:7d} floats ({p * 4:8d} B): {min(v):.3f} .. {max(v):.3f}")
