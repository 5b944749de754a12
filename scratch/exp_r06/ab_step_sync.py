#!/usr/bin/env python3
"""A/B on one box, interleaved processes: how `gpd_step_sync` waits -- the kernel's own completion word in page-locked memory (default)
against hipStreamSynchronize (GPD_STEP_SYNC_WAIT=stream, what rounds 6a-6i did) -- through the drop-in aviaries."""
import json
import os
import subprocess
import sys
import time

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, R)
    import warnings
    import numpy as np
    import torch
    from gym_pybullet_drones_amd.envs import HoverAviary, MultiHoverAviary
    from gym_pybullet_drones_amd.utils.enums import ActionType, Physics
    warnings.simplefilter("ignore")
    out = {}
    for name, make, A in (("HoverAviary()", lambda: HoverAviary(act=ActionType.ONE_D_RPM), (1, 1)),
                          ("HoverAviary(DYN, RPM, 240 Hz)", lambda: HoverAviary(physics=Physics.DYN, ctrl_freq=240), (1, 4)),
                          ("MultiHoverAviary(2, PID)", lambda: MultiHoverAviary(num_drones=2, physics=Physics.DYN, act=ActionType.PID), (2, 3))):
        env = make()
        acts = np.random.default_rng(0).uniform(-1, 1, size=(2420,) + A).astype(np.float32)
        env.reset(seed=0)
        for k in range(64):
            env.step(acts[k])
        env.reset(seed=0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(2420):
            _, _, term, trunc, _ = env.step(acts[k])
            if term or trunc:
                env.reset()
        dt = time.perf_counter() - t0
        core = env._core
        t1 = time.perf_counter()
        for _ in range(2000):
            core.step_host()
        t2 = time.perf_counter()
        out[name] = {"env_step_us": dt / 2420 * 1e6, "gpd_step_sync_us": (t2 - t1) / 2000 * 1e6}
        env.close()
    from gym_pybullet_drones_amd.control.DSLPIDControl import DSLPIDControl
    from gym_pybullet_drones_amd.utils.enums import DroneModel
    ctrl = DSLPIDControl(DroneModel.CF2X)
    rng = np.random.default_rng(1)
    pos, vel, tgt = rng.uniform(-1, 1, (2100, 3)), rng.uniform(-0.1, 0.1, (2100, 3)), rng.uniform(-1, 1, (2100, 3))
    quat = np.array([0.0, 0.0, 0.0, 1.0])
    for k in range(100):
        ctrl.computeControl(1 / 48, pos[k], quat, vel[k], np.zeros(3), tgt[k])
    t0 = time.perf_counter()
    for k in range(100, 2100):
        ctrl.computeControl(1 / 48, pos[k], quat, vel[k], np.zeros(3), tgt[k])
    out["DSLPIDControl.computeControl"] = {"env_step_us": (time.perf_counter() - t0) / 2000 * 1e6, "gpd_step_sync_us": float("nan")}
    print(json.dumps(out))
    raise SystemExit(0)

rows = {"word": [], "stream": []}
for rnd in range(4):
    for how in ("word", "stream"):
        env = dict(os.environ)
        env.pop("GPD_STEP_SYNC_WAIT", None)
        if how == "stream":
            env["GPD_STEP_SYNC_WAIT"] = "stream"
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, capture_output=True, text=True)
        line = next(l for l in r.stdout.splitlines() if l.startswith("{"))
        rows[how].append(json.loads(line))
        print(rnd, how, line, flush=True)
summary = {}
for name in rows["word"][0]:
    summary[name] = {how: {k: sorted(x[name][k] for x in rows[how])[len(rows[how]) // 2] for k in ("env_step_us", "gpd_step_sync_us")} for how in rows}
    summary[name]["saved_us_per_step"] = summary[name]["stream"]["env_step_us"] - summary[name]["word"]["env_step_us"]
print(json.dumps({"median_of_4_rounds": summary, "rounds": rows}))
os.makedirs(os.path.join(R, "gpurun_out", "r06m"), exist_ok=True)
json.dump({"median_of_4_rounds": summary, "rounds": rows}, open(os.path.join(R, "gpurun_out", "r06m", "ab_step_sync.json"), "w"), indent=1)
