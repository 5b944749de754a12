#!/usr/bin/env python3
"""Time the REFERENCE'S OWN Python on this container's host CPU (VERDICT r04 "next" #7).

Run in the build container only (needs /root/reference):   python scratch/time_reference_dyn.py

`pybullet` is not installable here, so the reference's default `Physics.PYB` cannot run; its own `Physics.DYN` path can: the
unmodified `HoverAviary` class of /root/reference, imported over `oracle/pybullet_shim.py` (a state store + the three restated
Bullet quaternion utilities -- the same stand-in `tests/golden/make_golden.py` uses to produce the golden fixtures).  The schedule
is BASELINE config 1's as SURVEY.md section 8(d) spells it out -- `HoverAviary()` defaults (30 Hz control, 240 Hz physics, S = 8),
ActionType.ONE_D_RPM, a ~ U(-1, 1) of shape (1, 1), 2 420 `env.step()` calls, reset when an episode ends (the loop of
examples/learn.py:54-58 / :157-192 without the policy) -- with `physics=Physics.DYN` instead of the default.

Writes profiles/r05_reference_python_dyn_cpu.json; `bench.py` quotes that file's figure next to its own port's (the file
travels to the GPU box, the reference does not).  Every arithmetic instruction timed here is the reference's; what the shim
replaces is PyBullet's state store (reset/get base pose + velocity) and three pure functions, all cheaper than the real calls."""
import json
import os
import platform
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(REPO, "tests", "golden"))
sys.path.insert(0, REPO)


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return platform.processor() or "unknown"


def schedule(HoverAviary, Physics, act, A, steps, repeats, **kw):
    runs = []
    for rep in range(repeats):
        env = HoverAviary(gui=False, physics=Physics.DYN, act=act, **kw)
        env.reset(seed=0)
        rng = np.random.default_rng(0)
        acts = rng.uniform(-1, 1, size=(steps, 1, A)).astype(np.float32)
        episodes = 0
        t0 = time.perf_counter()
        for i in range(steps):
            _, _, term, trunc, _ = env.step(acts[i])
            if term or trunc:
                env.reset()
                episodes += 1
        dt = time.perf_counter() - t0
        S = int(env.PYB_STEPS_PER_CTRL)
        env.close()
        runs.append({"seconds": dt, "env_steps_per_s": steps / dt, "drone_steps_per_s": steps * S / dt, "episodes_ended": episodes})
        print(f"run {rep}: {steps} env.step() in {dt:.2f} s = {steps * S / dt:.0f} drone-steps/s ({episodes} episodes ended)", flush=True)
    return runs


def main(steps=2420, repeats=3):
    import make_golden
    shim, mods = make_golden.load_reference()
    from gym_pybullet_drones.envs.HoverAviary import HoverAviary
    from gym_pybullet_drones.utils.enums import ActionType, Physics
    runs = schedule(HoverAviary, Physics, ActionType.ONE_D_RPM, 1, steps, repeats)
    best = max(runs, key=lambda r: r["drone_steps_per_s"])
    # the headline's own shape on the reference: 240 Hz control (one physics step per env.step(), so every step also pays the action
    # mapping, the observation and the task), ActionType.RPM -- what bench.py's `hover65536_240hz` runs per drone
    runs240 = schedule(HoverAviary, Physics, ActionType.RPM, 4, steps, repeats, pyb_freq=240, ctrl_freq=240)
    best240 = max(runs240, key=lambda r: r["drone_steps_per_s"])
    out = {
        "what": "the reference's own HoverAviary (unmodified, imported from /root/reference over oracle/pybullet_shim.py), Physics.DYN, "
                "ActionType.ONE_D_RPM, 30 Hz control / 240 Hz physics (S = 8), a ~ U(-1, 1), reset at episode end",
        "kind": "reference", "physics": "DYN", "cores": 1, "unit": "drone-steps/s",
        "value": best["drone_steps_per_s"], "env_steps_per_s": best["env_steps_per_s"],
        "steps": steps, "substeps_per_step": 8, "runs": runs,
        "at_240hz_control": {"what": "same class, pyb_freq = ctrl_freq = 240 (S = 1), ActionType.RPM: the headline workload's per-drone work",
                             "value": best240["drone_steps_per_s"], "unit": "drone-steps/s", "runs": runs240},
        "host_cpu": cpu_model(), "host_threads_online": os.cpu_count(), "python": platform.python_version(), "numpy": np.__version__,
        "where": "the build container (no GPU); the GPU box's host CPU is a different machine -- bench.py prints its own port on that box next to this figure",
        "not_timed": "Physics.PYB (Bullet's stepSimulation): pybullet is not installable in this image",
        "script": "scratch/time_reference_dyn.py",
    }
    path = os.path.join(REPO, "profiles", "r05_reference_python_dyn_cpu.json")
    json.dump(out, open(path, "w"), indent=1)
    print("wrote", path, "value", out["value"])


if __name__ == "__main__":
    main()
