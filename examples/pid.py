#!/usr/bin/env python3
"""Closed-loop DSLPID demo: a few Crazyflies chase each other around a circle.

The scenario of the reference's `examples/pid.py` (`CtrlAviary` stepped with RPMs that external
`DSLPIDControl` objects compute from the observed state, 48 Hz control / 240 Hz physics), twice:

  * `--mode dropin`  through the reference-shaped classes, one Python call per drone and step;
  * `--mode batched` E copies of the scene at once: `VectorCtrlAviary` + `DSLPIDControlBatch`, everything on the GPU.

Usage:  python examples/pid.py [--num_drones 3] [--duration_sec 3] [--mode dropin|batched] [--num_envs 1024]
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from gym_pybullet_drones_amd.control import DSLPIDControl, DSLPIDControlBatch  # noqa: E402
from gym_pybullet_drones_amd.envs import CtrlAviary, VectorCtrlAviary  # noqa: E402
from gym_pybullet_drones_amd.utils.enums import DroneModel, Physics  # noqa: E402
from gym_pybullet_drones_amd.utils.Logger import Logger  # noqa: E402
from gym_pybullet_drones_amd.utils.utils import str2bool  # noqa: E402


def circle_scene(num_drones, ctrl_hz, radius=0.3, height=0.1, height_step=0.05, period_sec=10):
    """Initial poses on a circle, the circular waypoint list and each drone's starting waypoint."""
    ang = np.arange(num_drones) / 6 * 2 * np.pi + np.pi / 2
    init_xyzs = np.stack([radius * np.cos(ang), radius * np.sin(ang) - radius, height + height_step * np.arange(num_drones)], 1)
    init_rpys = np.stack([np.zeros(num_drones), np.zeros(num_drones), np.arange(num_drones) * (np.pi / 2) / num_drones], 1)
    num_wp = ctrl_hz * period_sec
    th = np.arange(num_wp) / num_wp * 2 * np.pi + np.pi / 2
    waypoints = np.stack([radius * np.cos(th) + init_xyzs[0, 0], radius * np.sin(th) - radius + init_xyzs[0, 1]], 1)
    start = (np.arange(num_drones) * num_wp / 6).astype(int) % num_wp
    return init_xyzs, init_rpys, waypoints, start


def run(drone=DroneModel.CF2X, num_drones=3, physics=Physics.DYN, simulation_freq_hz=240, control_freq_hz=48, duration_sec=3,
        mode="dropin", num_envs=1024, output_folder="results", log=True, device="cuda:0"):
    init_xyzs, init_rpys, wps, wp = circle_scene(num_drones, control_freq_hz)
    steps = int(duration_sec * control_freq_hz)
    t0 = time.time()
    if mode == "dropin":
        env = CtrlAviary(drone_model=drone, num_drones=num_drones, initial_xyzs=init_xyzs, initial_rpys=init_rpys, physics=physics,
                         pyb_freq=simulation_freq_hz, ctrl_freq=control_freq_hz, device=device)
        ctrl = [DSLPIDControl(drone_model=drone, device=device) for _ in range(num_drones)]
        logger = Logger(logging_freq_hz=control_freq_hz, num_drones=num_drones, output_folder=output_folder) if log else None
        action = np.zeros((num_drones, 4))
        err = 0.0
        for i in range(steps):
            obs, reward, terminated, truncated, info = env.step(action)
            for j in range(num_drones):
                target = np.hstack([wps[wp[j]], init_xyzs[j, 2]])
                action[j, :], pos_e, _ = ctrl[j].computeControlFromState(control_timestep=env.CTRL_TIMESTEP, state=obs[j],
                                                                         target_pos=target, target_rpy=init_rpys[j])
                err = max(err, float(np.linalg.norm(pos_e))) if i > steps // 2 else err
                if logger:
                    logger.log(drone=j, timestamp=i / env.CTRL_FREQ, state=obs[j],
                               control=np.hstack([target, init_rpys[j], np.zeros(6)]))
            wp = (wp + 1) % len(wps)
        env.close()
        if logger:
            logger.save()
        final = obs[:, :3]
    else:
        E = num_envs
        env = VectorCtrlAviary(E, num_drones, drone_model=drone, initial_xyzs=init_xyzs, initial_rpys=init_rpys, physics=physics,
                               pyb_freq=simulation_freq_hz, ctrl_freq=control_freq_hz, device=device)
        ctrl = DSLPIDControlBatch(E * num_drones, drone, device=device)
        dev = env.device
        rpy_t = torch.as_tensor(np.tile(init_rpys, (E, 1)), dtype=torch.float32, device=dev)
        z_t = torch.as_tensor(np.tile(init_xyzs[:, 2], E), dtype=torch.float32, device=dev)
        wps_t = torch.as_tensor(wps, dtype=torch.float32, device=dev)
        wp_t = torch.as_tensor(np.tile(wp, E), device=dev)
        action = torch.zeros((E, num_drones, 4), device=dev)
        err = 0.0
        for i in range(steps):
            env.step(action)
            sv = env.state_vectors().view(E * num_drones, 20)
            target = torch.cat([wps_t[wp_t], z_t[:, None]], dim=1)
            rpm, pos_e, _ = ctrl.computeControl(env.CTRL_TIMESTEP, sv[:, 0:3], sv[:, 3:7], sv[:, 10:13], None, target, rpy_t)
            action = rpm.view(E, num_drones, 4)
            wp_t = (wp_t + 1) % len(wps)
            if i == steps - 1:
                err = float(pos_e.norm(dim=1).max())
        torch.cuda.synchronize()
        final = env.state_vectors().view(E, num_drones, 20)[0, :, :3].cpu().numpy()
    dt = time.time() - t0
    sim_steps = steps * (simulation_freq_hz // control_freq_hz) * num_drones * (num_envs if mode == "batched" else 1)
    print(f"[pid.py] mode={mode}: {steps} control steps in {dt:.2f}s wall ({sim_steps / dt:.3g} drone-steps/s), "
          f"tracking error (2nd half) {err:.3f} m, final xyz of aviary 0:\n{np.round(final, 3)}")
    return err, final


if __name__ == "__main__":
    ap = argparse.ArgumentParser(description="DSLPID circle-tracking demo (CtrlAviary + DSLPIDControl)")
    ap.add_argument("--drone", default=DroneModel.CF2X, type=DroneModel, choices=DroneModel)
    ap.add_argument("--num_drones", default=3, type=int)
    ap.add_argument("--physics", default=Physics.DYN, type=Physics, choices=Physics)
    ap.add_argument("--simulation_freq_hz", default=240, type=int)
    ap.add_argument("--control_freq_hz", default=48, type=int)
    ap.add_argument("--duration_sec", default=3, type=float)
    ap.add_argument("--mode", default="dropin", choices=["dropin", "batched"])
    ap.add_argument("--num_envs", default=1024, type=int)
    ap.add_argument("--output_folder", default="results")
    ap.add_argument("--log", default=True, type=str2bool)
    run(**vars(ap.parse_args()))
