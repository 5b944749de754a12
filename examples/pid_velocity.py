#!/usr/bin/env python3
"""Velocity-command demo: drones fly piecewise-constant velocity commands through `VelocityAviary`
(the scenario of the reference's `examples/pid_velocity.py`; the embedded DSLPID tracking runs inside the kernel).

Usage:  python examples/pid_velocity.py [--duration_sec 4]
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from gym_pybullet_drones_amd.envs import VelocityAviary  # noqa: E402
from gym_pybullet_drones_amd.utils.enums import DroneModel, Physics  # noqa: E402


def run(drone=DroneModel.CF2X, physics=Physics.DYN, simulation_freq_hz=240, control_freq_hz=48, duration_sec=4, device="cuda:0"):
    init_xyzs = np.array([[0, 0, .3], [.5, 0, .3], [1.0, 0, .3], [1.5, 0, .3]])
    init_rpys = np.array([[0, 0, 0], [0, 0, np.pi / 3], [0, 0, np.pi / 4], [0, 0, np.pi / 2]])
    env = VelocityAviary(drone_model=drone, num_drones=4, initial_xyzs=init_xyzs, initial_rpys=init_rpys, physics=physics,
                         pyb_freq=simulation_freq_hz, ctrl_freq=control_freq_hz, device=device)
    steps = int(duration_sec * control_freq_hz)
    # one command per drone: +x, +y, -x, up -- all at the speed limit
    cmd = np.array([[1, 0, 0, 1.0], [0, 1, 0, 1.0], [-1, 0, 0, 1.0], [0, 0, 1, 1.0]])
    obs, _ = env.reset()
    for i in range(steps):
        obs, reward, terminated, truncated, info = env.step(cmd)
    vel = obs[:, 10:13]
    print(f"[pid_velocity.py] after {duration_sec}s: velocities\n{np.round(vel, 3)}\n(speed limit {env.SPEED_LIMIT:.3f} m/s), "
          f"positions\n{np.round(obs[:, :3], 3)}")
    env.close()
    return vel, env.SPEED_LIMIT


if __name__ == "__main__":
    ap = argparse.ArgumentParser(description="Velocity-command demo (VelocityAviary)")
    ap.add_argument("--drone", default=DroneModel.CF2X, type=DroneModel, choices=DroneModel)
    ap.add_argument("--physics", default=Physics.DYN, type=Physics, choices=Physics)
    ap.add_argument("--simulation_freq_hz", default=240, type=int)
    ap.add_argument("--control_freq_hz", default=48, type=int)
    ap.add_argument("--duration_sec", default=4, type=float)
    run(**vars(ap.parse_args()))
