"""A/B of two builds of gpd_downwash_global on the swarm bench scene: dumps the force and the state after a few steps.
usage: GPD_LIB=<lib> python scratch/ab_swarm.py out.npz"""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
import bench

dev = torch.device("cuda:0")
env = bench.make_env(bench.WORKLOADS["swarm65536_ext_240hz"], dev, 0)
env.reset()
D = env.NUM_DRONES
rng = np.random.default_rng(1)
act = torch.as_tensor(env.HOVER_RPM * (1 + 0.005 * rng.uniform(-1, 1, size=(D, 4))).astype(np.float32), device=dev)
out = {}
out["f0"] = env.downwash().clone().cpu().numpy()
for k in range(40):
    env.step(act)
torch.cuda.synchronize()
out["dw"] = env.dw_force[:D].cpu().numpy()
out["kin"] = env.core.kin[:, :D].cpu().numpy()
np.savez(sys.argv[1], **{k: v for k, v in out.items() if v is not None})
print("nonzero forces:", int((out["dw"] != 0).sum()), "min", float(out["dw"].min()), "finite", bool(np.isfinite(out["kin"]).all()))
