#!/usr/bin/env python3
"""SURVEY.md §8(d) parity metric on the GPU box -> gpurun_out/parity_report.json (copied to profiles/rNN_parity_report.json: r01 with the first library, r04 with the final one).

65 536 HoverAviaries (cf2x, DYN, RPM actions), the HIP path (float32, `gpd_rollout` in launches of 64 steps and `gpd_step`)
against the C restatement of the oracle (float64, all host threads), identical fp32-rounded initial states and actions:
per field group g and time t, err_g(t) = ||x32 - x64||_inf / max(||x64||_inf over batch and time so far, floor_g) with floors
1 m, 1, 1 m/s, 1 rad/s; plus the share of state entries passing np.isclose(rtol=1e-4, atol=1e-5)."""
import json
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
from conftest import urdf                                     # noqa: E402
from oracle import c_oracle                                   # noqa: E402
from oracle.c_oracle import CAviary                           # noqa: E402
from test_gpu_parity import GROUPS, _core, _oracle_kin, _sync_from_oracle   # noqa: E402

dev = torch.device("cuda:0")
c_oracle.lib().orc_set_threads(os.cpu_count() or 1)
CHECK = (1, 10, 100, 240, 1920)
out = {"_doc": __doc__.strip(), "runs": {}}
for label, S, use_rollout in (("ctrl240_rollout64", 1, True), ("ctrl240_single_steps", 1, False), ("ctrl30_single_steps", 8, False)):
    rng = np.random.default_rng(65536)
    E, D = 65536, 1
    xyz = np.array([0, 0, 0.1125]) + rng.uniform(-0.5, 0.5, size=(E, D, 3)) * np.array([1, 1, 0])
    rpy = rng.uniform(-0.1, 0.1, size=(E, D, 3))
    orc = CAviary(urdf("cf2x"), "cf2x", E, D, initial_xyzs=xyz, initial_rpys=rpy, pyb_freq=240, ctrl_freq=240 // S, act="rpm",
                  task="hover")
    core = _core("cf2x", E, D, 0, S, "rpm", "hover", xyz, rpy, dev, target=orc.TARGET_POS)
    _sync_from_oracle(core, orc)
    n_env_steps = 1920 // S
    base = 0.01 * rng.uniform(-1, 1, size=(1, E, D, 4))
    maxima = {g: fl for g, (_, fl) in GROUPS.items()}
    rec = {}
    k = 0
    while k < n_env_steps:
        m = min(64, n_env_steps - k) if use_rollout else 1
        acts = (base + 0.01 * rng.uniform(-1, 1, size=(m, E, D, 4))).astype(np.float32)
        ta = torch.as_tensor(acts, device=dev)
        if use_rollout:
            core.rollout(ta)
        else:
            core.step(ta[0])
        for j in range(m):
            orc.step_in_place(acts[j].astype(np.float64))
            k += 1
            t = k * S
            if not use_rollout or j == m - 1 or t in CHECK:
                ref = _oracle_kin(orc)
                for g, (sl, _) in GROUPS.items():
                    maxima[g] = max(maxima[g], float(np.abs(ref[sl]).max()))
            if t in CHECK and (not use_rollout or j == m - 1):
                kin = core.kin[:, :E * D].cpu().numpy().astype(np.float64)
                rec[t] = {g: float(np.abs(kin[sl] - ref[sl]).max() / maxima[g]) for g, (sl, _) in GROUPS.items()}
                rec[t]["isclose_rtol1e-4_atol1e-5"] = float(np.isclose(kin, ref, rtol=1e-4, atol=1e-5).mean())
    if use_rollout:   # checkpoints that fall inside a 64-step launch are only reachable at launch ends: 1920 = 30 x 64
        pass
    out["runs"][label] = {"substeps": S, "env_steps": n_env_steps, "errors": rec}
    print(label, json.dumps(rec))
os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(REPO, "gpurun_out", "parity_report.json"), "w"), indent=1)
