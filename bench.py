#!/usr/bin/env python3
"""bench.py — throughput of the fused HIP env step on synthetic hover batches.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME] [--mode rollout|graph|eager]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one `env.step()` of EVERY aviary on this rank: the action of that step is read, the physics is
integrated, and the step's observation rows, rewards and terminated/truncated flags are written (same-step
auto-reset on).  Default workload `hover65536_240hz` (BASELINE.json's metric): 65 536 HoverAviaries per GPU (1 drone
each), Physics.DYN, ActionType.RPM, pyb_freq = ctrl_freq = 240 Hz (one physics step per env step, so env-steps ==
drone-steps); actions are pre-generated on the device and different every step.

Launch modes:
  rollout  (default, the headline `value`) `gpd_rollout`: 64 consecutive env steps per kernel launch -- the 64 action
           blocks are staged in HBM, every step's outputs are written, the drone state stays in registers;
  graph    one `gpd_step` launch per env step, 64 launches captured in a hipGraph (the pattern of an RL loop that
           runs a policy between steps); measured as well in the default run and reported under
           `one_launch_per_step`;
  eager    one host launch per step.
Weak scaling: every rank owns its own 65 536 aviaries; no data-path collective unless `--allgather` asks for the
optional RCCL all-gather of the observation shards.

Rank 0 prints ONE JSON line (metric/value/unit + roofline + cpu_baseline).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s achievable

WORKLOADS = {
    # name: (envs/GPU, drones/env, physics flags, ctrl_freq, act, task)
    "hover65536_240hz": dict(E=65536, D=1, phys=0, ctrl=240, act="rpm", task="hover"),
    "hover65536_30hz": dict(E=65536, D=1, phys=0, ctrl=30, act="rpm", task="hover"),
    "hover4096_240hz": dict(E=4096, D=1, phys=0, ctrl=240, act="rpm", task="hover"),
    "hover65536_ext_240hz": dict(E=65536, D=1, phys=7, ctrl=240, act="rpm", task="hover"),
    "stack8x8192_ext_240hz": dict(E=8192, D=8, phys=7, ctrl=240, act="rpm", task="multihover"),
    "multihover2x16384_240hz": dict(E=16384, D=2, phys=4, ctrl=240, act="rpm", task="multihover"),
    "hover65536_pid_240hz": dict(E=65536, D=1, phys=0, ctrl=240, act="pid", task="hover"),
    "hover4m_240hz": dict(E=4194304, D=1, phys=0, ctrl=240, act="rpm", task="hover"),
    "hover16m_240hz": dict(E=16777216, D=1, phys=0, ctrl=240, act="rpm", task="hover"),
    # ONE aviary of 65 536 drones, pairwise downwash over the whole swarm (gpd_downwash_global + gpd_step per sub-step)
    "swarm65536_ext_240hz": dict(E=1, D=65536, phys=7, ctrl=240, act="raw_rpm", task="none", swarm=True),
}


def make_env(w, device, seed):
    from gym_pybullet_drones_amd.envs import SwarmAviary, VectorAviary
    from gym_pybullet_drones_amd.utils.enums import ActionType, Physics
    E, D = w["E"], w["D"]
    rng = np.random.default_rng(seed)
    if w.get("swarm"):
        # 12 layers 1 m apart, a 4 m lattice per layer (74 x 74 sites) with +-0.3 m jitter: ~300 m x 300 m, 32 x 32 grid cells
        side = int(np.ceil(np.sqrt(D / 12)))
        idx = rng.permutation(side * side * 12)[:D]
        layer, site = idx // (side * side), idx % (side * side)
        xy = np.stack([(site % side) * 4.0, (site // side) * 4.0], axis=1) - 2.0 * side + rng.uniform(-0.3, 0.3, size=(D, 2))
        xyz = np.concatenate([xy, (1.0 + layer)[:, None]], axis=1)
        env = SwarmAviary(D, initial_xyzs=xyz, initial_rpys=rng.uniform(-0.05, 0.05, size=(D, 3)), physics=Physics.PYB_GND_DRAG_DW,
                          pyb_freq=240, ctrl_freq=w["ctrl"], act="raw_rpm", device=device)
        env.NUM_ENVS, env.ACT_DIM = 1, 4
        return env
    if D == 1:
        xyz = np.array([0, 0, 0.1125]) + rng.uniform(-0.5, 0.5, size=(E, D, 3)) * np.array([1, 1, 0])
    else:   # drones stacked 0.3 m apart so downwash / ground effect are active
        xyz = rng.uniform(-0.05, 0.05, size=(E, D, 3)) + np.arange(D)[None, :, None] * np.array([0.05, 0, 0.3]) + \
            np.array([0, 0, 0.1])
    rpy = rng.uniform(-0.1, 0.1, size=(E, D, 3))
    env = VectorAviary(E, D, initial_xyzs=xyz, initial_rpys=rpy, physics=w["phys"], pyb_freq=240,
                       ctrl_freq=w["ctrl"], act=ActionType(w["act"]), task=w["task"], auto_reset=True,
                       track_rpm=bool(w["phys"] & 2), device=device)
    return env


def make_actions(w, env, device, seed, pool):
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    a = torch.rand((pool, env.NUM_ENVS, env.NUM_DRONES, env.ACT_DIM), generator=g, device=device) * 2 - 1
    if w["act"] == "raw_rpm":
        a = float(env.HOVER_RPM) * (1 + 0.05 * a)
    if w["act"] == "pid":
        a = a * 0.5
        a[..., 2] += 1.0
    return a.contiguous()


def cpu_baseline(w, budget_s=12.0):
    """Time the loop-structured float64 oracle (the CPU 'port' of the reference's per-drone Python/numpy
    path; PyBullet itself is not installable here) on ONE host core, on a bounded sample of the workload."""
    from oracle.aviary_oracle import OracleAviary
    urdf = os.path.join(REPO, "gym-pybullet-drones_amd", "assets", "cf2x.urdf")
    D = w["D"]
    env = OracleAviary(urdf, "cf2x", num_drones=D, physics_flags=w["phys"], pyb_freq=240, ctrl_freq=w["ctrl"],
                       act=w["act"], task=w["task"] if w["task"] != "hover" or D == 1 else "multihover")
    rng = np.random.default_rng(0)
    A = env.action_buffer[0].shape[1]
    acts = rng.uniform(-1, 1, size=(64, D, A))
    env.step(acts[0])
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < budget_s:
        for k in range(16):
            _, _, term, trunc = env.step(acts[(n + k) % 64])
            if term or trunc:
                env.reset()
        n += 16
    dt = time.perf_counter() - t0
    S = 240 // w["ctrl"]
    out = {"value": n * D * S / dt, "unit": "drone-steps/s", "cores": 1, "kind": "port",
           "sample": f"{n} env.step() of ONE aviary ({D} drone(s), S={S}) through oracle/aviary_oracle.py "
                     f"(float64 per-drone numpy loop restating BaseAviary._dynamics + BaseRLAviary + task) in {dt:.1f}s "
                     f"on 1 host core; PyBullet (Physics.PYB) is not installable in this image"}
    try:    # second figure: the same arithmetic compiled (oracle/gpd_oracle.c, scalar float64, one core)
        from oracle.c_oracle import CAviary
        Ec = 2048
        c = CAviary(urdf, "cf2x", Ec, D, physics_flags=w["phys"], pyb_freq=240, ctrl_freq=w["ctrl"], act=w["act"],
                    task=w["task"] if w["task"] != "hover" or D == 1 else "multihover")
        ac = rng.uniform(-1, 1, size=(8, Ec, D, A))
        c.step_in_place(ac[0])
        m, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < 3.0:
            c.step_in_place(ac[m % 8])
            m += 1
        dtc = time.perf_counter() - t0
        out["c_port"] = {"value": m * Ec * D * S / dtc, "unit": "drone-steps/s", "cores": 1,
                         "sample": f"{m} steps of {Ec} aviaries through oracle/gpd_oracle.c (gcc -O2, scalar float64) in {dtc:.1f}s"}
        # third figure: the same C restatement with the aviaries spread over every host core (OpenMP)
        from oracle import c_oracle
        threads = c_oracle.lib().orc_set_threads(os.cpu_count() or 1)
        try:
            Ea = 2048 * max(1, min(threads, 64) // 2)
            ca = CAviary(urdf, "cf2x", Ea, D, physics_flags=w["phys"], pyb_freq=240, ctrl_freq=w["ctrl"], act=w["act"],
                         task=w["task"] if w["task"] != "hover" or D == 1 else "multihover")
            aa = rng.uniform(-1, 1, size=(4, Ea, D, A))
            ca.step_in_place(aa[0])
            m, t0 = 0, time.perf_counter()
            while time.perf_counter() - t0 < 3.0:
                ca.step_in_place(aa[m % 4])
                m += 1
            dta = time.perf_counter() - t0
            out["c_port_all_cores"] = {"value": m * Ea * D * S / dta, "unit": "drone-steps/s", "cores": threads,
                                       "sample": f"{m} steps of {Ea} aviaries, OpenMP over aviaries, {threads} threads, in {dta:.1f}s"}
        finally:
            c_oracle.lib().orc_set_threads(1)
    except Exception as e:   # the C restatement is optional test infrastructure
        out["c_port"] = {"error": str(e)[:200]}
    return out


def measure(mode, args, env, actions, gather, device, world, POOL):
    """Time K env steps of every aviary on this rank.  mode 'graph': one kernel launch per env step, 64
    steps captured in a hipGraph; 'eager': one host launch per step; 'rollout': `gpd_rollout`, POOL steps
    per launch (actions of the POOL steps pre-staged, every step's obs/reward/flags written)."""
    from gym_pybullet_drones_amd import dist as gdist
    core = env.core

    def one_step(i):
        env.step(actions[i % POOL])
        if gather is not None:
            gather(core.obs12)

    # exactly K timed steps: K // POOL full groups (a POOL-step rollout / a replay of the POOL-step graph) plus one
    # group of K % POOL steps (a shorter rollout / a second, shorter graph)
    K, W = args.steps, args.warmup
    rem = K % POOL if mode != "eager" else 0
    if mode != "eager":
        W = (W + POOL - 1) // POOL * POOL          # (untimed warm-up: whole groups only, reported as run)

    gathers = {}
    if mode == "rollout" and gather is not None:
        for n in {POOL, rem} - {0}:
            gathers[n] = gdist.ObsAllGather(n * core.N, 12, device=device)   # one larger collective per rollout

    def one_rollout(n):
        obs = core.rollout(actions[:n], update_latest=False)[0]
        if n in gathers:
            gathers[n](obs.view(-1, 12))

    graphs = {}
    if mode == "rollout":
        one_rollout(POOL)
        if rem:
            one_rollout(rem)
    else:
        for i in range(min(max(W, 1), 64)):
            one_step(i)
    torch.cuda.synchronize()
    if mode == "graph":
        for n in {POOL, rem} - {0}:
            stream = torch.cuda.Stream(device)
            stream.wait_stream(torch.cuda.current_stream(device))
            with torch.cuda.stream(stream):
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=stream):
                    for i in range(n):
                        one_step(i)
            torch.cuda.current_stream(device).wait_stream(stream)
            graphs[n] = g

    def run(k):
        if mode == "eager":
            for i in range(k):
                one_step(i)
            return
        for n in [POOL] * (k // POOL) + ([k % POOL] if k % POOL else []):
            if mode == "rollout":
                one_rollout(n)
            elif n in graphs:
                graphs[n].replay()
            else:                       # (a warm-up remainder the timed region does not need a graph for)
                for i in range(n):
                    one_step(i)

    run(W)
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev0.record()          # on the current stream = the stream every gpd_* launch above goes to
    run(K)
    ev1.record()
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    t1 = time.perf_counter()
    wall = gdist.max_over_ranks(t1 - t0, device=device)
    ev_ms = ev0.elapsed_time(ev1)
    # roofline of the dominant kernel: algorithmic bytes of all launches of the timed region / its HIP-event time
    if mode == "rollout":
        launches = K // POOL + (1 if rem else 0)
        bytes_total = (K // POOL) * core.bytes_per_rollout(POOL) + (core.bytes_per_rollout(rem) if rem else 0)
    else:
        launches, bytes_total = K, K * core.bytes_per_step()
    bytes_launch = bytes_total / launches
    launch_us = ev_ms * 1e3 / launches
    achieved = bytes_total / (ev_ms * 1e-3) / 1e9
    n_total = core.N * world
    steps_per_launch = K / launches
    return {
        "K": K, "W": W, "wall": wall, "value": n_total * core.S * K / wall, "env_steps_per_s": n_total * K / wall,
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                     "kernel": ("gpd_rollout1_kernel" if core.D == 1 else "gpd_rollout_kernel") if mode == "rollout"
                     else "gpd_step_kernel",
                     "env_steps_per_launch": steps_per_launch, "bytes_per_launch": bytes_launch,
                     "bytes_per_drone_per_env_step": bytes_launch / (core.N * steps_per_launch),
                     "launch_us_hip_events": launch_us},
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=32768)
    ap.add_argument("--warmup", type=int, default=2048)
    ap.add_argument("--workload", default="hover65536_240hz", choices=sorted(WORKLOADS))
    ap.add_argument("--mode", default="rollout", choices=["rollout", "graph", "eager"],
                    help="rollout: gpd_rollout, 64 env steps per launch (state in registers, actions pre-staged); "
                         "graph: one launch per env step, hipGraph of 64 steps; eager: one host launch per step")
    ap.add_argument("--allgather", action="store_true", help="all-gather the obs shards over RCCL")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-second-leg", action="store_true",
                    help="skip the extra one-launch-per-step measurement reported next to the rollout headline")
    args = ap.parse_args()

    from gym_pybullet_drones_amd import dist as gdist
    # (GPD_DIST_BACKEND / GPD_BENCH_SINGLE_DEVICE: test hooks -- run the multi-rank code path with gloo on one GPU)
    rank, world, local = gdist.init_from_env(os.environ.get("GPD_DIST_BACKEND", "nccl") if args.gpus > 1 else None)
    if args.gpus != world:
        if rank == 0:
            print(f"[bench] --gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run", file=sys.stderr)
        args.gpus = world
    assert torch.cuda.is_available(), "bench.py needs a GPU (the product path has no CPU fallback)"
    device = torch.device("cuda", local if world > 1 and not os.environ.get("GPD_BENCH_SINGLE_DEVICE") else 0)
    torch.cuda.set_device(device)

    w = WORKLOADS[args.workload]
    env = make_env(w, device, seed=1000 + rank)
    POOL = 64      # env steps per rollout launch / per captured hipGraph
    actions = make_actions(w, env, device, seed=2000 + rank, pool=POOL)
    core = env.core
    gather = gdist.ObsAllGather(core.N, 12, device=device) if args.allgather else None

    if w.get("swarm") and args.mode == "rollout":
        args.mode = "graph"          # a single world needs the downwash of every sub-step's snapshot: one step per launch group
    second = None
    if args.mode == "rollout" and not args.no_second_leg:
        second = measure("graph", args, env, actions, gather, device, world, POOL)
        env.reset()
    m = measure(args.mode, args, env, actions, gather, device, world, POOL)

    if rank == 0:
        n_total = core.N * world
        launch = {"rollout": f"rollout{POOL}", "graph": "graph", "eager": "eager"}[args.mode]
        out = {
            "metric": "env steps/sec (whole node), HoverAviary N=65536 drones @240Hz",
            "value": m["value"], "unit": "drone-steps/s", "n_gpus": world, "steps": m["K"], "warmup": m["W"],
            "ms_per_step": m["wall"] * 1e3 / m["K"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": args.workload, "envs_per_gpu": core.E, "drones_per_env": core.D,
                       "total_drones": n_total, "physics": "DYN" + "".join(n for b, n in ((1, "+GND"), (2, "+DRAG"), (4, "+DW")) if w["phys"] & b),
                       "pyb_freq": 240, "ctrl_freq": w["ctrl"], "substeps_per_step": core.S, "action": w["act"],
                       "task": w["task"], "auto_reset": True, "launch": launch,
                       "obs_allgather": bool(args.allgather), "env_steps_per_s": m["env_steps_per_s"]},
            "roofline": m["roofline"],
        }
        if second is not None:
            out["one_launch_per_step"] = {"value": second["value"], "unit": "drone-steps/s", "steps": second["K"],
                                          "launch": "graph (hipGraph of 64 single-step launches)",
                                          "roofline": second["roofline"]}
        tfile = os.path.join(REPO, "profiles", "hbm_traffic.json")
        if os.path.exists(tfile):   # measured offline in separate --pmc passes (see the file's _comment)
            table = json.load(open(tfile))
            for key, roof in ((f"{args.workload}:{launch}", out["roofline"]),
                              (f"{args.workload}:graph", second["roofline"] if second else None)):
                rec = table.get(key)
                if rec and roof is not None and rec.get("traffic_bytes"):
                    roof["traffic"] = rec["traffic_bytes"]
                    roof["traffic_source"] = "profiles/hbm_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE)"
                    roof["rocprof_kernel_avg_us"] = rec["rocprof_kernel_avg_ns"] / 1e3
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(w)
        print(json.dumps(out))
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
