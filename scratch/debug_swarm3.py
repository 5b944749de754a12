#!/usr/bin/env python3
"""Round 3: the persistent swarm path on the bench scene -- per step: dmax^2 (the meta row), wall time of the step (synchronised),
rows without a finite position; for two re-binning schedules."""
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402

dev = torch.device("cuda:0")
w = bench.WORKLOADS["swarm65536_ext_240hz"]
for cell, rebin in ((10.5, 16),):
    os.environ["GPD_SWARM_CELL"], os.environ["GPD_SWARM_REBIN"] = str(cell), str(rebin)
    env = bench.make_env(w, dev, seed=1000)
    acts = bench.make_actions(w, env, dev, seed=2000, pool=64)
    env.reset()
    torch.cuda.synchronize()
    N = env.NUM_DRONES
    print(f"cell {cell} rebin {rebin}: grid {env.nx} x {env.ny}, rows {env.n_rows}")
    for k in range(256):
        t0 = time.perf_counter()
        env.step(acts[k % 64])
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        meta = env.pos4[N:, 3].max().cpu().numpy()
        fin = int(env.pos4[:N, :3].isfinite().all(dim=1).sum())
        bx = env._bin_pos[:N, :3]
        d2 = ((env.pos4[:N, :3] - bx) ** 2).sum(dim=1)
        v = env.core.kin[7:10, :N]
        sp = (v ** 2).sum(dim=0).sqrt()
        top = torch.topk(sp, 3)
        if k % 8 == 7 or float(d2.max()) > 0.0625:
            i = int(top.indices[0])
            print(f"     fastest drones {top.indices.tolist()} speeds {[round(x, 2) for x in top.values.tolist()]}; drone {i}: pos {env.core.kin[0:3, i].tolist()} F {float(env.dw_force[i]):.3f}")
        if k % 8 == 7 or float(d2.max()) > 0.0625:
          print(f"  step {k:2d}: {dt * 1e6:8.1f} us  meta {meta}  finite {fin}  true dmax^2 {float(d2.max()):.3e}  since_bin {env._since_bin}  |F|max {float(env.dw_force[:N].abs().max()):.3f}")
