#!/usr/bin/env python3
"""Throughput demo: E HoverAviaries stepped (a) one launch per step with a stand-in policy between steps and
(b) K steps per launch with `rollout()` on a pre-computed action sequence; through the SB3-shaped adapter last.

Usage:  python examples/rollout.py [--num_envs 65536] [--steps 256]
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from gym_pybullet_drones_amd.envs import VecEnvAdapter, VectorHoverAviary  # noqa: E402
from gym_pybullet_drones_amd.utils.enums import ActionType  # noqa: E402


def run(num_envs=65536, steps=256, device="cuda:0"):
    env = VectorHoverAviary(num_envs, act=ActionType.ONE_D_RPM, ctrl_freq=30, full_obs=True, device=device)
    obs, _ = env.reset()
    w = torch.randn((env.OBS_DIM, 1), device=env.device) * 0.01            # a stand-in linear "policy"
    torch.cuda.synchronize()
    t0 = time.time()
    ret = torch.zeros(num_envs, device=env.device)
    for _ in range(steps):
        action = torch.tanh(obs.view(num_envs, -1) @ w).view(num_envs, 1, 1)
        obs, reward, terminated, truncated, info = env.step(action)
        ret += reward
    torch.cuda.synchronize()
    t1 = time.time()
    acts = torch.rand((steps, num_envs, 1, 1), device=env.device) * 2 - 1
    torch.cuda.synchronize()
    t2 = time.time()
    obs_k, rew_k, term_k, trunc_k = env.rollout(acts)
    torch.cuda.synchronize()
    t3 = time.time()
    S = env.PYB_STEPS_PER_CTRL
    print(f"[rollout.py] {num_envs} aviaries x {steps} env steps (x{S} physics sub-steps): policy-in-the-loop "
          f"{num_envs * steps * S / (t1 - t0):.3g} drone-steps/s, rollout {num_envs * steps * S / (t3 - t2):.3g} drone-steps/s; "
          f"mean return {float(ret.mean()):.2f}, episodes ended in the rollout: {int((term_k | trunc_k).sum())}")
    venv = VecEnvAdapter(VectorHoverAviary(64, act=ActionType.ONE_D_RPM, ctrl_freq=30, full_obs=True, device=device), squeeze=True)
    o = venv.reset()
    for _ in range(8):
        o, r, d, infos = venv.step(np.zeros((64, 1), dtype=np.float32))
    print(f"[rollout.py] VecEnvAdapter: obs {o.shape} {o.dtype}, rewards {r.shape}, dones {int(d.sum())}")
    return obs_k.shape


if __name__ == "__main__":
    ap = argparse.ArgumentParser(description="Vector env stepping vs rollout")
    ap.add_argument("--num_envs", default=65536, type=int)
    ap.add_argument("--steps", default=256, type=int)
    run(**vars(ap.parse_args()))
