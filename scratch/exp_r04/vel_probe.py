"""the VEL variant of gpd_rollout_policy under the DEFAULT instruction scheduler (GPD_LIB=scratch/exp_r04/
libgpd_defsched.so: the main unit compiled without -amdgpu-sched-strategy=max-ilp).  What exactly comes back wrong?"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gym_pybullet_drones_amd.envs import VectorHoverAviary
from gym_pybullet_drones_amd.policy import MlpPolicy
from gym_pybullet_drones_amd.utils.enums import ActionType
from gym_pybullet_drones_amd import _native
print("library:", os.environ.get("GPD_LIB", _native.LIB_PATH))
dev = torch.device("cuda:0")
for act, hist in ((ActionType.VEL, False), (ActionType.VEL, True), (ActionType.PID, False), (ActionType.ONE_D_PID, False)):
    for task_on in (True, False):
        E, K, ctrl = 300, 12, 30
        mk = lambda: VectorHoverAviary(E, act=act, ctrl_freq=ctrl, full_obs="lazy" if hist else False, device=dev, initial_xyzs=np.array([[0, 0, 0.5]]))
        a, b = mk(), mk()
        if not task_on:
            for e in (a, b):
                e.core._cfg.task = 0
        A, H = a.ACT_DIM, ctrl // 2
        pol = MlpPolicy.random(12 + (H * A if hist else 0), A, seed=3, gain=1.2, device=dev)
        obs, rew, term, trunc, acts = a.rollout_policy(pol, K)
        first_bad = None
        for t in range(K):
            o, r, te, tr, _ = b.step(acts[t])
            same = (torch.equal(o, obs[t]), torch.equal(r, rew[t]), torch.equal(te, term[t]), torch.equal(tr, trunc[t]))
            if not all(same) and first_bad is None:
                first_bad = (t, same, float(rew[t].mean()), float(r.mean()), float(trunc[t].float().mean()), float(tr.float().mean()),
                             float(term[t].float().mean()), float(te.float().mean()))
        print(f"{act.name:10s} hist={hist!s:5s} task={'hover' if task_on else 'none '}:", "bitwise equal to gpd_step" if first_bad is None else
              f"FIRST MISMATCH step {first_bad[0]} (obs, rew, term, trunc equal: {first_bad[1]}); mean reward policy {first_bad[2]:.4f} vs step {first_bad[3]:.4f}; "
              f"truncated frac {first_bad[4]:.3f} vs {first_bad[5]:.3f}; terminated frac {first_bad[6]:.3f} vs {first_bad[7]:.3f}")
