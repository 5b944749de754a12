#!/usr/bin/env python3
"""What the same-step auto-reset costs the headline rollout: 65 536 HoverAviaries at 240 Hz, K steps per launch, with the bench's U(-1, 1)
RPM actions (1 % of the drones end an episode per step: about half of all wave-steps run the episode-end block) against zero actions
(hover: nobody resets), HIP events over 200 launches each, interleaved."""
import json
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from gym_pybullet_drones_amd.envs import VectorAviary
from gym_pybullet_drones_amd.utils.enums import ActionType

dev = torch.device("cuda", 0)
out = {}
for K in (20, 64):
    env = VectorAviary(65536, 1, physics=0, pyb_freq=240, ctrl_freq=240, act=ActionType("rpm"), task="hover", auto_reset=True, device=dev)
    acts = {"random": (torch.rand((K, 65536, 1, 4), device=dev) * 2 - 1).contiguous(), "hover": torch.zeros((K, 65536, 1, 4), device=dev),
            "small": ((torch.rand((K, 65536, 1, 4), device=dev) * 2 - 1) * 0.02).contiguous()}
    for rnd in range(3):
        for name, a in acts.items():
            env.core.reset()
            for _ in range(5):
                env.core.rollout(a, update_latest=False)
            t = bench.event_seconds(lambda: env.core.rollout(a, update_latest=False), 200) * 1e6
            _, _, te, tr = env.core.rollout(a, update_latest=False)
            ended = float((te | tr).float().mean())
            out.setdefault(f"K{K}_{name}", []).append({"us_per_launch": t, "us_per_step": t / K, "episodes_ended_per_drone_step": ended})
            print(K, name, round(t, 3), round(t / K, 4), ended, flush=True)
os.makedirs("gpurun_out/r06x", exist_ok=True)
json.dump(out, open("gpurun_out/r06x/reset_cost.json", "w"), indent=1)
