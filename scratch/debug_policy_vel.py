import os, sys
import numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
from test_gpu_policy import _env
from gym_pybullet_drones_amd.policy import MlpPolicy
from gym_pybullet_drones_amd.utils.enums import Physics
dev = torch.device("cuda:0")
for name, kw in [("default", {}), ("xy huge", dict(xy_bound=1e9)), ("z huge", dict(z_bound=1e9)), ("tilt huge", dict(tilt_bound=1e9)), ("counter huge", dict(trunc_counter=2**30)),
                 ("all bounds huge", dict(xy_bound=1e9, z_bound=1e9, tilt_bound=1e9)), ("task none", dict(task=0))]:
    E, K = 777, 2
    a = _env("vel", 240, False, E, dev, episode_len_sec=2.0, physics=Physics("dyn"), auto_reset=False)
    for k, v in kw.items():
        setattr(a.core._cfg, k, v)
    pol = MlpPolicy.random(12, a.ACT_DIM, seed=3, gain=1.2, device=dev)
    obs, rew, term, trunc, acts = a.rollout_policy(pol, K)
    print(f"{name:16s} term {term.sum(1).tolist()} trunc {trunc.sum(1).tolist()} rew {rew[0][:2].tolist()}")
