#!/usr/bin/env python3
"""VERDICT r05 #5, candidate 2: the three per-step small stores of the headline kernel (reward 4 B + two flag bytes per aviary) folded into
ONE 8-byte record (`gpd_rollout_packed`), unpacked host-side as strided views.  A/B against `gpd_rollout` on the driver's command shape
(65 536 HoverAviaries, 240 Hz, K = 20 steps per launch) and at K = 64, interleaved rounds in one process, HIP events; bitwise check first.
Result (profiles/r06_ab_packed_step_records.json): +0.67 % / +0.06 % -- not kept.  The variant is scratch/exp_r06/packed_records.patch
(`patch -p1` on the tree, rebuild, then this script runs)."""
import json
import os
import sys

import numpy as np
import torch

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import bench  # noqa: E402

dev = torch.device("cuda:0")
w = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "hover65536_240hz"]
res = {}
for K in (20, 64):
    a, b = bench.make_env(w, dev, seed=1000), bench.make_env(w, dev, seed=1000)
    acts = bench.make_actions(w, a, dev, seed=2000, pool=K)
    same = True
    for _ in range(6):          # 6 launches: through resets and episode ends
        oa = a.core.rollout(acts, update_latest=False)
        ob = b.core.rollout_packed(acts)
        same = same and all(torch.equal(x, y) for x, y in zip(oa, ob)) and torch.equal(a.core.kin_store, b.core.kin_store)
    assert same, "packed records differ from gpd_rollout"

    def timed(fn, reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / reps          # us per launch
    reps = 6000 if K == 20 else 3000
    rounds = {"rollout": [], "packed": []}
    for r in range(7):
        rounds["rollout"].append(timed(lambda: a.core.rollout(acts, update_latest=False), reps))
        rounds["packed"].append(timed(lambda: b.core.rollout_packed(acts), reps))
    med = {k: float(np.median(v)) for k, v in rounds.items()}
    res[f"K{K}"] = {"bitwise_equal": same, "us_per_launch": rounds, "median_us_per_launch": med, "us_per_step": {k: v / K for k, v in med.items()},
                    "gain": med["rollout"] / med["packed"] - 1.0}
    print(K, json.dumps(res[f"K{K}"]), flush=True)
os.makedirs(os.path.join(R, "gpurun_out"), exist_ok=True)
json.dump(res, open(os.path.join(R, "gpurun_out", "ab_packed.json"), "w"), indent=1)
