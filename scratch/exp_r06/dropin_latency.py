#!/usr/bin/env python3
"""The drop-in single-aviary step: `HoverAviary()` with its defaults, ActionType.ONE_D_RPM, 2 420 steps -- BASELINE config
1's shape -- wall clock per env.step(), with the aviary's state in host-visible memory (the default since round 6) and in HBM
(GPD_HOST_VISIBLE=0: the round-5 path), and where the time of a step goes."""
import json
import os
import sys
import time

import numpy as np

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import torch  # noqa: E402
from gym_pybullet_drones_amd.envs import HoverAviary, MultiHoverAviary  # noqa: E402
from gym_pybullet_drones_amd.utils.enums import ActionType, Physics  # noqa: E402

res = {}
for label, hv in (("host_visible", "1"), ("hbm_state", "0")):
    os.environ["GPD_HOST_VISIBLE"] = hv
    for name, make, A in (("HoverAviary()", lambda: HoverAviary(act=ActionType.ONE_D_RPM), (1, 1)),
                          ("HoverAviary(DYN, RPM)", lambda: HoverAviary(physics=Physics.DYN), (1, 4)),
                          ("MultiHoverAviary(2)", lambda: MultiHoverAviary(num_drones=2, physics=Physics.DYN), (2, 4))):
        env = make()
        rng = np.random.default_rng(0)
        acts = rng.uniform(-1, 1, size=(2420,) + A).astype(np.float32)
        env.reset(seed=0)
        for k in range(50):
            env.step(acts[k])
        env.reset(seed=0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eps = 0
        for k in range(2420):
            _, _, term, trunc, _ = env.step(acts[k])
            if term or trunc:
                env.reset()
                eps += 1
        dt = time.perf_counter() - t0
        # where it goes: the launch + drain alone, then the numpy refresh alone
        core = env._core
        a = core.action_host if core.host_visible else torch.zeros((core.N, core.A), device=core.device)
        t1 = time.perf_counter()
        for k in range(1000):
            core.step(a)
            if not core.host_visible:
                torch.cuda.synchronize()
        t2 = time.perf_counter()
        for k in range(1000):
            env._updateAndStoreKinematicInformation()
        t3 = time.perf_counter()
        for k in range(1000):
            env._computeObs()
        t4 = time.perf_counter()
        res[f"{label}: {name}"] = {"us_per_step": dt / 2420 * 1e6, "episodes_ended": eps, "launch_and_drain_us": (t2 - t1) * 1e3,
                                   "kinematic_refresh_us": (t3 - t2) * 1e3, "compute_obs_us": (t4 - t3) * 1e3}
        print(label, name, json.dumps(res[f"{label}: {name}"]), flush=True)
        env.close()
os.makedirs(os.path.join(R, "gpurun_out"), exist_ok=True)
json.dump(res, open(os.path.join(R, "gpurun_out", "dropin_latency.json"), "w"), indent=1)
