#!/usr/bin/env python3
"""bench_extra.py — the workloads beside the hot path's headline, through bench.py's own harness (same flags, same JSON line):

  * ONE aviary of any size (`swarm65536_ext_240hz`, `swarm1m_ext_240hz`): pairwise downwash over the whole world, `--gpus N` shares the
    world among N ranks (strong scaling: halo exchange or all-gather of positions per sub-step);
  * the policy inside the rollout kernel (`*_policy*`, gpd_rollout_policy) -- frozen since round 4;
  * materialised history rows / the action ring pushed in the kernel (`*_fullobs`, `*_history`).

    python bench_extra.py --workload swarm65536_ext_240hz [--gpus N] [--steps K] [--warmup W] ...

The driver runs bench.py; nothing here is part of the headline.  The losing experiments of rounds 3 - 5 (`--split`, `--stagger`,
`--cu-mask`, `--rollout-graph`) are not carried: `git show d596b96:bench.py` has their last form, profiles/HISTORY.md their results.
"""
import json
import os
import sys

import numpy as np
import torch

import bench
from bench import NUM_SIMDS, PEAK_CLOCK_GHZ, REPO

# ---------------------------------------------------------------------------------------------------------------------------------
# one world of any size
# ---------------------------------------------------------------------------------------------------------------------------------


def swarm_env(w, device, seed, E=None, world=1, rank=0, job=None):
    from gym_pybullet_drones_amd import dist as gdist
    from gym_pybullet_drones_amd.envs import SwarmAviary
    from gym_pybullet_drones_amd.utils.enums import Physics
    D = w["D"]
    rng = np.random.default_rng(1000)          # ONE world: the same scene on every rank, rank r takes its block of drones
    note, exch = [], None
    if world > 1:
        # GPD_SWARM_EXCHANGE=halo (default): blocks from the neighbouring stripes only (gpd_p2p_group: grouped ncclSend / ncclRecv;
        # torch.distributed's batched P2P where the native communicator is missing); =allgather: every position to every rank
        from gym_pybullet_drones_amd.envs import NativeHaloExchange, NativeSlabExchange, TorchHaloExchange, TorchSlabExchange
        kind = os.environ.get("GPD_SWARM_EXCHANGE", "halo")
        margin = float(os.environ.get("GPD_SWARM_HALO_MARGIN", "2.0"))
        native, torch_ = (NativeHaloExchange, TorchHaloExchange) if kind == "halo" else (NativeSlabExchange, TorchSlabExchange)
        err = None
        try:
            exch = (native(margin=margin, device=device) if kind == "halo" else native(device=device)) if job.backend == "nccl" else None
        except Exception as e:      # noqa: BLE001
            err = f"{type(e).__name__}: {e}"[:200]
        if not gdist.all_ranks_ok(exch is not None, device=device):
            exch = torch_(margin=margin) if kind == "halo" else torch_()
            note.append(f"position exchange ({kind}) through torch.distributed ({err or 'no native RCCL communicator'})")
        else:
            note.append(f"position exchange ({kind}): " + ("gpd_p2p_group, grouped ncclSend / ncclRecv" if kind == "halo" else "gpd_allgather_obs in place") +
                        f", {exch.nc.ranks_seen} ranks seen by RCCL")
    # 12 layers 1 m apart, a 4 m lattice per layer with +-0.1 m jitter.  Layer l is shifted by (l % 4, l // 4) metres inside the lattice
    # cell, so no drone hovers within 0.8 m (laterally) of one above it: with drones stacked vertically the reference's downwash model
    # pushes the lower one down, it falls onto the next one, and alpha ~ 1/dz^2 diverges as they pass
    side = int(np.ceil(np.sqrt(D / 12)))
    idx = rng.permutation(side * side * 12)[:D]
    layer, site = idx // (side * side), idx % (side * side)
    xy = np.stack([(site % side) * 4.0 + layer % 4, (site // side) * 4.0 + layer // 4], axis=1) - 2.0 * side + rng.uniform(-0.1, 0.1, size=(D, 2))
    xyz = np.concatenate([xy, (1.0 + layer)[:, None]], axis=1)
    kw = {k: v for k, v in (("cell", os.environ.get("GPD_SWARM_CELL")), ("rebin_every", os.environ.get("GPD_SWARM_REBIN"))) if v}
    env = SwarmAviary(D, initial_xyzs=xyz, initial_rpys=rng.uniform(-0.05, 0.05, size=(D, 3)), physics=Physics.PYB_GND_DRAG_DW,
                      pyb_like="damped", pyb_freq=240, ctrl_freq=w["ctrl"], act="raw_rpm", device=device, world_size=world, rank=rank, exchange=exch,
                      cell=float(kw.get("cell", 10.5)), rebin_every=int(kw["rebin_every"]) if "rebin_every" in kw else None)
    env.NUM_ENVS, env.ACT_DIM = 1, 4
    # a single world has no task and no auto-reset: every pass of the schedule starts from the initial lattice (one reset launch per pass)
    env.reset_each_pass = True
    env.bench_kernel = dict.fromkeys(("graph", "eager"), "dwg_force_kernel (+ gpd_swarm_step_kernel; a binning every few sub-steps)")
    env.bench_note = note
    return env


def swarm_mode(mode, world, backend):
    if mode == "rollout":
        mode = "graph"          # a single world needs the downwash of every sub-step's snapshot: one step per launch group
    if world > 1 and backend == "gloo" and mode == "graph":
        mode = "eager"          # (the gloo test hook stages the position exchange through host memory: not capturable)
    return mode


def swarm_parity(args, job, env):
    if args.no_parity or not env.flags & 4:
        return None
    all_pos = env.all_positions() if job.world > 1 else None          # (collective: a rank of a halo-exchanging world holds its neighbourhood only)
    if job.rank != 0:
        return None
    try:
        from oracle.bench_checks import swarm_parity_check
        return swarm_parity_check(env, all_pos)
    except Exception as e:          # noqa: BLE001
        return {"error": f"{type(e).__name__}: {e}"[:300]}


def swarm_pairs(env):
    """Pairs the wake lists of the current binning hold = what every replay launch evaluates.  None without lists."""
    if getattr(env, "_pair_list", None) is None:
        return None
    torch.cuda.synchronize()
    nb = env._pair_nb[..., 0].to(torch.int32) & 0xffff                         # [groups, 4] batches per wave
    ok = env._list_ok.to(torch.bool)
    cap64 = env._pair_list.shape[2]
    idx = torch.arange(cap64, device=env.device).view(1, 1, -1)
    live = (idx < (nb * 64).unsqueeze(2)) & ok.view(-1, 1, 1)
    pairs = int(((env._pair_list != -1) & live).sum().item())
    slots = int((nb * 64)[ok].sum().item())
    return {"pairs": pairs, "list_slots": slots, "groups_with_a_list": int(ok.sum().item()), "groups": int(ok.numel()),
            "list_bytes_read_per_substep": slots * 4}


def swarm_finish(out, args, job, env, m, clock_ghz):
    """The one-world blocks of the line: the world's configuration, the roofline that binds it (VALU issue: wave-instructions of all
    kernels of a sub-step x 4 cycles on 1024 SIMDs against the measured sub-step), its CPU baseline.  Collective part first."""
    halo_check = None
    if job.world > 1 and getattr(env.exchange, "halo", False):
        try:        # (collective) did every drone of every rank stay within the halo's margin since the last plan?
            env.exchange.check(env)
            halo_check = "ok"
        except RuntimeError as e:
            halo_check = str(e)[:300]
    if out is None:
        return
    world = job.world
    out["config"]["total_drones"] = env.TOTAL_DRONES
    out["config"]["swarm"] = {"total_drones": env.TOTAL_DRONES, "ranks": env.WORLD_SIZE, "cell_m": env.cell, "grid": [env.nx, env.ny],
                              "rebin_every": env.rebin_every, "note": "; ".join(env.bench_note) or None,
                              "exchange": None if world == 1 else ("halo" if getattr(env.exchange, "halo", False) else "allgather"),
                              "exchange_bytes_sent_per_rank_per_substep": None if world == 1 else
                              (env.exchange.bytes_per_substep if getattr(env.exchange, "halo", False) else env.slab * 16),
                              "exchange_bytes_received_allgather": None if world == 1 else (world - 1) * env.slab * 16,
                              "halo_plans_made": getattr(env.exchange, "plans_made", None), "halo_margin_check": halo_check}
    roof = out["roofline"]
    hbm = {k: roof[k] for k in ("bound", "achieved", "peak", "unit", "frac", "achievable", "frac_of_achievable", "bytes_per_launch",
                                "bytes_per_drone_per_env_step", "floor_us") if k in roof}
    try:
        pairs = swarm_pairs(env)
        us = m["us_per_step"] / max(env.PYB_STEPS_PER_CTRL, 1)                 # per physics sub-step
        rec = None
        f = os.path.join(REPO, "profiles", "swarm_counters.json")
        if os.path.exists(f):
            rec = json.load(open(f)).get(out["config"]["workload"])
        peak = NUM_SIMDS * PEAK_CLOCK_GHZ / 4.0                                # G wave-instructions per second
        new = {"bound": "valu_issue", "achieved": None, "peak": peak, "unit": "G wave-instructions/s", "frac": None, "traffic": roof.get("traffic"),
               "kernel": roof["kernel"], "us_per_substep": us, "pairs": pairs, "hbm": hbm, "clock": roof.get("clock"),
               "env_steps_per_launch": roof.get("env_steps_per_launch"), "launch_us_hip_events": roof.get("launch_us_hip_events")}
        if pairs:
            new.update(pairs_per_substep=pairs["pairs"], pairs_per_drone=pairs["pairs"] / env.TOTAL_DRONES, pair_evaluations_per_s=pairs["pairs"] / (us * 1e-6))
        if rec:
            valu = rec["valu_wave_instructions_per_substep"]
            new["achieved"] = valu / (us * 1e-6) / 1e9
            new["frac"] = new["achieved"] / peak
            new["valu_floor_us"] = valu * 4.0 / (NUM_SIMDS * PEAK_CLOCK_GHZ * 1e3)
            if clock_ghz:
                new["frac_at_measured_clock"] = valu * 4.0 / (NUM_SIMDS * clock_ghz * 1e3) / us
            new["counters"] = rec
            new["traffic"] = rec.get("hbm_bytes_per_substep")
            if pairs and rec.get("replay_valu_wave_instructions"):
                new["valu_lane_instructions_per_pair"] = rec["replay_valu_wave_instructions"] * 64.0 / pairs["pairs"]
            new["source"] = "profiles/swarm_counters.json (rocprofv3 --pmc SQ_INSTS_*, scratch/profile_r05.py)"
        else:
            new["note"] = "no profiles/swarm_counters.json entry for this workload: instruction counts unknown, frac not computed"
        out["roofline"] = new
    except Exception as e:      # noqa: BLE001 -- the line survives without the extra block
        out["roofline"]["swarm_roofline_error"] = f"{type(e).__name__}: {e}"[:200]
    if not args.no_cpu_baseline and world == 1:
        from oracle.bench_checks import swarm_cpu_baseline
        out["cpu_baseline"] = swarm_cpu_baseline(bench.WORKLOADS[args.workload], env)


SWARM = dict(phys=7, ctrl=240, act="raw_rpm", task="none", swarm=True, scaling="strong", builder=swarm_env, pick_mode=swarm_mode,
             parity=swarm_parity, finish=swarm_finish)
bench.WORKLOADS["swarm65536_ext_240hz"] = dict(SWARM, E=1, D=65536)
bench.WORKLOADS["swarm1m_ext_240hz"] = dict(SWARM, E=1, D=1048576)

# ---------------------------------------------------------------------------------------------------------------------------------
# history rows, the policy in the kernel
# ---------------------------------------------------------------------------------------------------------------------------------


def rows_env(w, device, seed, E=None, world=1, rank=0, job=None):
    plain = {k: v for k, v in w.items() if k not in ("builder", "parity", "finish", "pick_mode")}
    env = bench.make_env(plain, device, seed, E=E, world=world, rank=rank)
    c = env.core
    hist = getattr(env, "full_obs", False) or getattr(env, "lazy_history", False)
    pol = None
    if w.get("policy"):
        from gym_pybullet_drones_amd.policy import MlpPolicy
        pol = MlpPolicy.random(12 + (env.ACTION_BUFFER_SIZE * env.ACT_DIM if w.get("full_obs") else 0), env.ACT_DIM, seed=seed, gain=1.0, device=device)
        noise = mean = None
        if w.get("sample"):
            noise = torch.randn((bench.POOL, c.N, env.ACT_DIM), device=device)
            mean = torch.empty_like(noise)

        def launch(a, n):           # the policy IN the kernel (gpd_rollout_policy); with noise rows: PPO's collection form
            if noise is not None:
                return c.rollout_policy(pol, n, want_actions=True, noise=noise[:n], action_std=[0.6] * env.ACT_DIM, mean_out=mean[:n])
            return c.rollout_policy(pol, n, want_actions=True)

        def step(actions, i):       # the policy between two steps, as torch operations on the rows gathered for it
            env.step(pol(env.full_rows() if pol.in_dim > 12 else c.obs12.view(-1, 1, 12)))
        env.bench_launch, env.bench_step = launch, step
        env.bench_kernel = {"rollout": "gpd_rollout_policy_kernel"}
    elif hist:
        env.bench_launch = lambda a, n: env.rollout(a[:n])

    def extra(n, rollout):          # materialised (12 + H*A)-float rows / the ring update behind a rollout
        if pol is not None:         # in the kernel: the ring push; between steps: the gathered rows (+ the MLP's own traffic, not counted)
            ring = 2 * n * c.N * c.A * 4 if getattr(env, "lazy_history", False) else 0
            return ring if rollout else (c.bytes_full_rows(n) if pol.in_dim > 12 else 0)
        if getattr(env, "full_obs", False):
            return c.bytes_full_rows(n, push=rollout)
        if getattr(env, "lazy_history", False) and rollout:
            if getattr(c, "pushed_history", False):     # gpd_rollout_history: the action (already counted) goes to both ring halves
                return n * c.N * 2 * c.A * 4 + 2 * 4 * c.E
            return c.bytes_full_rows(n, push=True) - c.bytes_full_rows(n)     # post-pass (gpd_full_obs, ring update only)
        return 0
    env.bench_extra_bytes = extra
    return env


def rows_finish(out, args, job, env, m, clock_ghz):
    if out is None:
        return
    w = bench.WORKLOADS[args.workload]
    out["config"]["policy"] = "MlpPolicy 64x64 tanh, in the kernel (rollout) / torch between steps (graph)" if w.get("policy") else None
    if not args.no_cpu_baseline and job.world == 1:
        from oracle.bench_checks import cpu_baseline
        out["cpu_baseline"] = cpu_baseline(w, phys=env.core.physics_flags)


ROWS = dict(E=65536, D=1, phys=0, act="rpm", task="hover", builder=rows_env, finish=rows_finish)
NO_REPLAY = dict(parity=lambda args, job, env: None)          # (the replay checker drives plain action blocks: not a policy's)
bench.WORKLOADS.update({
    "hover65536_240hz_fullobs": dict(ROWS, ctrl=240, full_obs=True),
    "hover65536_30hz_fullobs": dict(ROWS, ctrl=30, full_obs=True),
    # ... and with the action ring only ("lazy": the history tail stays a strided view of the ring the step kernel pushes into)
    "hover65536_240hz_history": dict(ROWS, ctrl=240, full_obs="lazy"),
    "hover65536_30hz_history": dict(ROWS, ctrl=30, full_obs="lazy"),
    # the loop of examples/learn.py:157-192 with the policy IN the kernel (SB3's default 2 x 64 tanh actor on the matrix cores, the policy
    # sees the reference's full 72-float row); second leg: the same policy as torch operations between two gpd_step launches
    "hover65536_30hz_policy": dict(ROWS, **NO_REPLAY, ctrl=30, full_obs="lazy", policy=True),
    "hover65536_240hz_policy12": dict(ROWS, **NO_REPLAY, ctrl=240, policy=True),
    "hover65536_30hz_policy_sample": dict(ROWS, **NO_REPLAY, ctrl=30, full_obs="lazy", policy=True, sample=True),
})

if __name__ == "__main__":
    bench.__file__ = os.path.abspath(__file__)        # (a self-launched multi-rank job starts THIS file on every rank)
    bench.main(sys.argv[1:])
