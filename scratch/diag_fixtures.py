import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, 'tests')
import numpy as np, torch
from conftest import golden
from gym_pybullet_drones_amd.envs import MultiHoverAviary, CtrlAviary
from gym_pybullet_drones_amd.control import DSLPIDControl
from gym_pybullet_drones_amd.utils.enums import ActionType, Physics, DroneModel
dev = torch.device('cuda:0')
g = golden('multihover_pid')
env = MultiHoverAviary(num_drones=3, physics=Physics.DYN, act=ActionType.PID, device=dev)
env.reset()
for k, a in enumerate(g['actions'][:60]):
    obs, rew, term, trunc, _ = env.step(a)
    sv = np.array([env._getDroneStateVector(i) for i in range(3)])
    e = np.abs(sv - g['state20'][k])
    print('mh_pid', k, 'pos %.2e quat %.2e vel %.2e angv %.2e rpm %.2e' % (e[:, :3].max(), e[:, 3:7].max(), e[:, 10:13].max(), e[:, 13:16].max(), e[:, 16:].max()), 'rew', rew, float(g['reward'][k]))
g = golden('ctrl_pid_circle_cf2x')
n, hz = 3, 48
env = CtrlAviary(drone_model=DroneModel.CF2X, num_drones=n, initial_xyzs=g['init_xyzs'], initial_rpys=g['init_rpys'], physics=Physics.DYN, pyb_freq=240, ctrl_freq=hz, device=dev)
ctrl = [DSLPIDControl(DroneModel.CF2X, device=dev) for _ in range(n)]
action = np.zeros((n, 4))
for k in range(144):
    obs, *_ = env.step(action)
    e = np.abs(obs - g['obs'][k])
    for j in range(n):
        action[j], _, _ = ctrl[j].computeControlFromState(env.CTRL_TIMESTEP, obs[j], g['target'][k, j], g['init_rpys'][j])
    er = np.abs(action - g['rpm'][k]).max()
    if k % 4 == 0 or k < 8: print('circle', k, 'pos %.2e quat %.2e vel %.2e angv %.2e | rpm err %.2e' % (e[:, :3].max(), e[:, 3:7].max(), e[:, 10:13].max(), e[:, 13:16].max(), er))
