#!/usr/bin/env python3
"""Does ONE 8-byte record per aviary and step (instead of a reward array and two flag arrays: three write streams -> one) change what
HBM delivers at 4 194 304 drones?  Needs the tree patched with scratch/exp_r06/packed_records.patch and rebuilt (gpu_call8.sh does that
on the GPU box).  Per trial: a fresh environment, its blocks placed by the arena search (gpd_rollout), then gpd_rollout_packed on the
SAME observation / action blocks with the record block freshly allocated three times (best kept)."""
import gc
import json
import os
import sys

import numpy as np
import torch

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import bench  # noqa: E402
from gym_pybullet_drones_amd.placement import place_rollout  # noqa: E402

dev = torch.device("cuda:0")
w = bench.WORKLOADS["hover4m_240hz"]
K = 64
rows = []
for trial in range(int(sys.argv[1]) if len(sys.argv) > 1 else 5):
    env = bench.make_env(w, dev, seed=1000 + trial)
    core = env.core
    arena, rep = place_rollout(core, K, target=0.755)
    acts = arena.actions
    plain = core.bytes_per_rollout(K) / bench.event_seconds(lambda: core.rollout(acts, update_latest=False), 20) / 8e12
    obs = core._rollout_cache[K][0]
    packed = []
    for _ in range(3):
        rec = torch.zeros((K, core.E), dtype=torch.int64, device=dev)
        by = rec.view(torch.uint8).view(K, core.E, 8)
        core.__dict__["_packed_cache"] = {K: (obs, rec, rec.view(torch.float32).view(K, core.E, 2)[..., 0], by[..., 4].view(torch.bool), by[..., 5].view(torch.bool))}
        core.rollout_packed(acts)
        # (8 bytes per aviary-step instead of 6: the algorithmic bytes of the packed launch)
        nbytes = core.bytes_per_rollout(K) + 2 * K * core.E
        packed.append(nbytes / bench.event_seconds(lambda: core.rollout_packed(acts), 20) / 8e12)
    row = {"trial": trial, "rollout_frac": plain, "search": {k: rep[k] for k in ("probes", "seen")}, "packed_frac_three_record_blocks": packed}
    rows.append(row)
    print(json.dumps(row), flush=True)
    del env, core, arena, acts, obs, rec, by
    gc.collect()
    torch.cuda.empty_cache()
json.dump(rows, open(os.path.join(R, "gpurun_out", "ab_packed_hbm.json"), "w"), indent=1)
