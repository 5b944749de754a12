#!/usr/bin/env python3
"""What the pair-balanced placement can buy on the bench's one-world scene (GPU box): the groups' batch counts after a build, the work
per CU (workgroup b -> CU b mod 256) under the identity placement, the device's placement and an offline greedy (longest-processing-time)
placement, and the replay launch timed under each of them with the permutation frozen.  -> gpurun_out/balance_diag_r05.txt"""
import ctypes
import os
import sys

import numpy as np
import torch

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import bench  # noqa: E402

dev = torch.device("cuda:0")
out = []


def say(*a):
    line = " ".join(str(x) for x in a)
    print(line, flush=True)
    out.append(line)


def cu_load(w_slot):
    return np.bincount(np.arange(len(w_slot)) % 256, weights=w_slot, minlength=256)


def lpt(w_group):
    """greedy: groups by weight descending, each to the least loaded CU that still has one of its four first-round slots free"""
    G = len(w_group)
    order = np.argsort(-w_group, kind="stable")
    load, used = np.zeros(256), np.zeros(256, dtype=int)
    perm = np.empty(G, dtype=np.int32)
    extra = 1024
    for g in order:
        free = np.where(used < 4)[0]
        if len(free) and extra <= G:
            c = free[np.argmin(load[free])]
            perm[used[c] * 256 + c] = g
            used[c] += 1
            load[c] += w_group[g]
        else:
            perm[extra] = g
            extra += 1
    return perm


def run(name, freeze_perm, steps=400):
    w = bench.WORKLOADS["swarm65536_ext_240hz"]
    os.environ["GPD_SWARM_REBIN"] = "100000"
    env = bench.make_env(w, dev, seed=1000)
    G = (env.n_rows + 63) // 64
    if freeze_perm is not None:
        p = torch.as_tensor(np.concatenate([freeze_perm, freeze_perm, [2]]).astype(np.int32), device=dev)
        env._group_perm.copy_(p)
    env.reset()                                         # bin + build under the permutation in force
    rpm = torch.full((env.NUM_DRONES, 4), float(env.HOVER_RPM), device=dev)
    for _ in range(8):
        env.step(rpm)
    torch.cuda.synchronize()
    # the replay launch alone, back to back (positions do not change between calls: the same pairs every time)
    L, sw, prm = env.core.lib, env._sw, env.core._params
    st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(50):
        L.gpd_swarm_forces(ctypes.byref(prm), ctypes.byref(sw), 0, st)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(steps):
        L.gpd_swarm_forces(ctypes.byref(prm), ctypes.byref(sw), 0, st)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / steps
    nb = (env._pair_nb.cpu().numpy().astype(np.int64)[:, :, 0] & 0xffff)
    perm = env._group_perm.cpu().numpy()
    cur = int(perm[2 * G]) & 1
    p = perm[cur * G:(cur + 1) * G]
    w_slot = nb.sum(axis=1)
    w_group = np.zeros(G, dtype=np.int64)
    w_group[p] = w_slot
    c = cu_load(w_slot)
    say(f"{name:10s} replay launch {us:6.2f} us (eager, back to back) | batches per CU: max {c.max():.0f} mean {c.mean():.1f} min {c.min():.0f} | heaviest wave {nb.max()} "
        f"| list ok {float(env._list_ok.float().mean()):.3f} | is a permutation: {sorted(p.tolist()) == list(range(G))}")
    return w_group, p


ident = None
wg, p = run("device", None)
G = len(wg)
say("groups", G, "batches per group: mean %.1f max %d p99 %.0f" % (wg.mean(), wg.max(), np.percentile(wg, 99)))
ci = cu_load(wg)
say("identity placement would give: max %.0f mean %.1f" % (ci.max(), ci.mean()))
run("identity", np.arange(G))
pl = lpt(wg.astype(float))
say("offline greedy placement gives: max %.0f" % cu_load(wg[pl]).max())
run("greedy", pl)
open(os.path.join(R, "gpurun_out", "balance_diag_r05.txt"), "w").write("\n".join(out) + "\n")
