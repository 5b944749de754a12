import sys, numpy as np, torch
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from conftest import golden
from gym_pybullet_drones_amd.envs import VelocityAviary
from gym_pybullet_drones_amd.utils.enums import DroneModel, Physics
g = golden("velocity_aviary_cf2x")
n, hz = g["init_xyzs"].shape[0], int(g["ctrl_hz"])
env = VelocityAviary(drone_model=DroneModel.CF2X, num_drones=n, initial_xyzs=g["init_xyzs"], initial_rpys=g["init_rpys"], physics=Physics.DYN, pyb_freq=240, ctrl_freq=hz, device="cuda:0")
env.reset()
for k in range(60):
    obs, *_ = env.step(g["actions"][k])
    ref = g["obs"][k]
    e = np.abs(obs[:, :16] - ref[:, :16])
    print(k, "pos %.2e quat %.2e rpy %.2e vel %.2e angv %.2e rpm %.2e" % (e[:, 0:3].max(), e[:, 3:7].max(), e[:, 7:10].max(), e[:, 10:13].max(), e[:, 13:16].max(), np.abs(obs[:, 16:] - ref[:, 16:]).max()))
