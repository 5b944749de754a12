#!/usr/bin/env python3
"""bench.py — throughput of the fused HIP env step on synthetic hover batches (the driver-run file).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME] [--mode rollout|graph|eager]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one `env.step()` of EVERY aviary on this rank: the action of that step is read, the physics is integrated, and the
step's observation rows, rewards and terminated/truncated flags are written (same-step auto-reset on).  Default workload
`hover65536_240hz` (BASELINE.json's metric): 65 536 HoverAviaries per GPU (1 drone each), Physics.DYN, ActionType.RPM,
pyb_freq = ctrl_freq = 240 Hz (env-steps == drone-steps); actions are pre-generated on the device and different every step.

Launch modes: `rollout` (default, the headline `value`) = `gpd_rollout`, up to 64 consecutive env steps per
launch, action blocks staged in HBM, every step's outputs written, the state in registers; `graph` = one `gpd_step` launch per env
step, up to 64 launches in a hipGraph (an RL loop with a policy between steps; reported under `one_launch_per_step` as well);
`eager` = one host launch per step (`python_step_loop`).

What is timed: the K steps of `--steps` form one SCHEDULE (K // 64 launches of 64 steps + one of K % 64), repeated back to back
until the timed region lasts >= `--min-time`; `ms_per_step` / `value` / `roofline.achieved` come from ONE clock, HIP events on the
launch stream (max over ranks); the host wall clock around the same region is printed beside it.  Weak scaling: every rank owns its
aviaries, no data-path collective unless a workload names the optional RCCL all-gather of the observation shards.
The line also carries `hbm_saturating`, `dropin_single_env`, `parity`, `cpu_baseline` and, for N > 1, the `suite` (BASELINE configs
4 and 5).  Checker code: oracle/bench_checks.py; one-world, policy and history-row workloads: bench_extra.py.  ONE JSON line, rank 0.
"""
import argparse
import ctypes
import json
import math
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
HBM_ACHIEVABLE_GBS = 6300.0    # ... and what a streaming kernel reaches on it (same guide)
PEAK_CLOCK_GHZ = 2.4       # MI355X peak engine clock
NUM_SIMDS = 256 * 4        # 256 CUs x 4 SIMDs; a wave64 VALU instruction occupies its SIMD for 4 cycles
BASELINE_METRIC = "env steps/sec (whole node), HoverAviary N=65536 drones @240Hz"
POOL = 64                  # env steps per rollout launch / per captured hipGraph

WORKLOADS = {
    # name: envs/GPU, drones/env, physics flags, ctrl_freq, act, task
    "hover65536_240hz": dict(E=65536, D=1, phys=0, ctrl=240, act="rpm", task="hover"),
    "hover65536_30hz": dict(E=65536, D=1, phys=0, ctrl=30, act="rpm", task="hover"),
    "hover4096_240hz": dict(E=4096, D=1, phys=0, ctrl=240, act="rpm", task="hover"),
    "hover65536_ext_240hz": dict(E=65536, D=1, phys=7, ctrl=240, act="rpm", task="hover"),
    # the headline with terminal observations kept (what VecEnvAdapter / GymVectorEnvAdapter need): the store-wave kernel
    "hover65536_240hz_termobs": dict(E=65536, D=1, phys=0, ctrl=240, act="rpm", task="hover", term_obs=True),
    "stack8x8192_ext_240hz": dict(E=8192, D=8, phys=7, ctrl=240, act="rpm", task="multihover"),
    "multihover2x16384_240hz": dict(E=16384, D=2, phys=4, ctrl=240, act="rpm", task="multihover"),
    "hover65536_pid_240hz": dict(E=65536, D=1, phys=0, ctrl=240, act="pid", task="hover"),
    # config 2 at the reference's 30 Hz control, and the closed-loop (ActionType.PID) run of every config
    "hover4096_30hz": dict(E=4096, D=1, phys=0, ctrl=30, act="rpm", task="hover"),
    "hover4096_pid_240hz": dict(E=4096, D=1, phys=0, ctrl=240, act="pid", task="hover"),
    "hover65536_ext_pid_240hz": dict(E=65536, D=1, phys=7, ctrl=240, act="pid", task="hover"),
    "stack8x8192_ext_pid_240hz": dict(E=8192, D=8, phys=7, ctrl=240, act="pid", task="multihover"),
    "multihover2x16384_pid_240hz": dict(E=16384, D=2, phys=4, ctrl=240, act="pid", task="multihover"),
    "hover4m_240hz": dict(E=4194304, D=1, phys=0, ctrl=240, act="rpm", task="hover"),
    "hover16m_240hz": dict(E=16777216, D=1, phys=0, ctrl=240, act="rpm", task="hover"),
    # BASELINE.json configs 4 and 5, per GPU, verbatim (launch with --gpus 8)
    "hover65536x8_allgather": dict(E=65536, D=1, phys=0, ctrl=240, act="rpm", task="hover", allgather=True),
    # (config 5's two drones are stacked 0.3 m apart like every multi-drone workload here: from MultiHoverAviary's DEFAULT poses --
    # both drones at z = 0.1125 -- the reference's downwash model, alpha ~ 1/dz^2, returns ~1e10 N as soon as rounding separates
    # the heights; no trajectory from that start means anything, in any precision)
    "multihover2x16384x8": dict(E=16384, D=2, phys=4, ctrl=240, act="rpm", task="multihover"),
}
# (bench_extra.py adds workloads with a "builder": one world of any size, the policy in the kernel, materialised history rows)


def stack_scene(rng, E, D):
    """Initial poses of the multi-drone workloads: D drones stacked so that downwash and ground effect are active, the lowest
    0.1 m above the ground, 0.3 m apart -- or, taller stacks, as far apart as keeps the top drone under the task's 2 m ceiling.
    THE scene of BASELINE config 3 (ii) and 5: `tests/test_gpu_fullsize.py` checks the same poses it comes from here.
      D == 2: the pair 5 cm apart laterally (the upper drone's wake pushes the lower one with ~2x its weight, it falls away);
      D  > 2: a staircase, 12 cm per drone (a drone sits on the shoulder of its upper neighbour's wake, |beta| ~ 0.07 m, where
              the force is ~0.2x its weight), +-0.05 rad of tilt: the well-conditioned start SURVEY.md section 8(d) asks for."""
    dz = min(0.3, 1.7 / max(D - 1, 1))
    if D == 2:
        xyz = rng.uniform(-0.05, 0.05, size=(E, D, 3)) + np.arange(D)[None, :, None] * np.array([0.05, 0, dz]) + np.array([0, 0, 0.1])
        return xyz, rng.uniform(-0.1, 0.1, size=(E, D, 3))
    xyz = rng.uniform(-0.02, 0.02, size=(E, D, 3)) + np.arange(D)[None, :, None] * np.array([0.12, 0, dz]) + np.array([0, 0, 0.1])
    return xyz, rng.uniform(-0.05, 0.05, size=(E, D, 3))


def make_env(w, device, seed, E=None, world=1, rank=0, job=None):
    if w.get("builder"):                      # (bench_extra.py's workloads build their own environment)
        return w["builder"](w, device, seed, E=E, world=world, rank=rank, job=job)
    from gym_pybullet_drones_amd.envs import VectorAviary
    from gym_pybullet_drones_amd.utils.enums import ActionType
    E, D = E or w["E"], w["D"]
    rng = np.random.default_rng(seed)
    if D == 1:
        xyz = np.array([0, 0, 0.1125]) + rng.uniform(-0.5, 0.5, size=(E, D, 3)) * np.array([1, 1, 0])
        rpy = rng.uniform(-0.1, 0.1, size=(E, D, 3))
    else:
        xyz, rpy = stack_scene(rng, E, D)
    return VectorAviary(E, D, initial_xyzs=xyz, initial_rpys=rpy, physics=w["phys"], pyb_freq=240, ctrl_freq=w["ctrl"],
                        act=ActionType(w["act"]), task=w["task"], auto_reset=True, track_rpm=bool(w["phys"] & 2),
                        full_obs=w.get("full_obs", False), keep_terminal_obs=bool(w.get("term_obs")), device=device)


def make_actions(w, env, device, seed, pool):
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    a = torch.rand((pool, env.NUM_ENVS, env.NUM_DRONES, env.ACT_DIM), generator=g, device=device) * 2 - 1
    if w["act"] == "raw_rpm":
        a = float(env.HOVER_RPM) * (1 + 0.005 * a)
    if w["act"] == "pid":
        a = a * 0.5
        a[..., 2] += 1.0
    return a.contiguous()


def host_threads():
    """Threads this process may really use: the affinity mask, capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(math.ceil(int(quota) / int(period)))))
    except (OSError, ValueError):
        pass
    return n


def groups_of(k, pool):
    return [pool] * (k // pool) + ([k % pool] if k % pool else [])


def launch_rollout(e, a, n):
    """One rollout launch of n env steps of the aviaries of `e` with the action blocks a[:n] -- THE timed call of rollout mode (also
    what the parity leg replays).  Returns (obs12 [n,N,12], reward, terminated, truncated[, actions])."""
    hook = getattr(e, "bench_launch", None)          # (bench_extra.py: policy rollouts, history rows)
    if hook is not None:
        return hook(a, n)
    return e.core.rollout(a[:n], update_latest=False)


def measure(mode, args, env, actions, gather, device, world):
    """Time `repeats` x K env steps of every aviary on this rank.  mode 'graph': one kernel launch per env step, up to POOL steps
    captured in a hipGraph; 'eager': one host launch per step; 'rollout': `gpd_rollout`, up to POOL steps per launch (actions
    pre-staged, every step's obs/reward/flags written)."""
    from gym_pybullet_drones_amd import dist as gdist
    core = env.core
    K, W = args.steps, args.warmup
    main = torch.cuda.current_stream(device)
    step_hook = getattr(env, "bench_step", None)

    def one_step(i):
        if step_hook is not None:
            step_hook(actions, i)
        else:
            env.step(actions[i % POOL])
        if gather is not None:
            gather(core.obs12)

    sizes = set(groups_of(K, POOL)) | set(groups_of(W, POOL))
    gathers = {}
    if mode == "rollout" and gather is not None:
        for n in sorted(sizes):             # one larger collective per rollout; every length shares the ONE communicator of `gather`
            gathers[n] = gather.sized(n * core.N * 12)

    def one_rollout(n):
        out = launch_rollout(env, actions, n)
        if n in gathers:
            gathers[n](out[0].reshape(-1, 12))

    graphs = {}
    if mode == "rollout":
        for n in sorted(sizes, reverse=True):
            one_rollout(n)                          # (allocates the per-length output buffers outside the timed region)
    else:
        for i in range(min(max(W, 1), 8)):
            one_step(i)
    torch.cuda.synchronize()
    if mode == "graph":
        for n in set(groups_of(K, POOL)):
            stream = torch.cuda.Stream(device)
            stream.wait_stream(main)
            with torch.cuda.stream(stream):
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=stream):
                    for i in range(n):
                        one_step(i)
            main.wait_stream(stream)
            graphs[n] = g

    def run(k):
        if getattr(env, "reset_each_pass", False):
            env.reset()
        if mode == "eager":
            for i in range(k):
                one_step(i)
            return
        for n in groups_of(k, POOL):
            if mode == "rollout":
                one_rollout(n)
            elif n in graphs:
                graphs[n].replay()
            else:                       # (a warm-up remainder the timed region does not need a graph for)
                for i in range(n):
                    one_step(i)

    per_rank_s, marks = [], []          # marks: (repeats done, seconds since the first event) at the segment events of the last timed() call

    def timed(reps):
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ev0.record()          # on the current stream = the stream every gpd_* launch above goes to
        marks.clear()
        nseg = int(getattr(args, "segment_events", 0) or 0)
        every = max(1, reps // nseg) if nseg else 0
        for i in range(reps):
            run(K)
            if every and (i + 1) % every == 0 and i + 1 < reps:        # the timed region seen piecewise (clock ramps, throttling)
                ev = torch.cuda.Event(enable_timing=True)
                ev.record()
                marks.append((i + 1, ev))
        ev1.record()
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        t1 = time.perf_counter()
        mine = ev0.elapsed_time(ev1) * 1e-3
        marks[:] = [(n, ev0.elapsed_time(ev) * 1e-3) for n, ev in marks] + [(reps, mine)]
        per_rank_s[:] = gdist.gather_floats(mine, device=device)       # (this rank's own event time, from every rank)
        return gdist.max_over_ranks(mine, device=device), gdist.max_over_ranks(t1 - t0, device=device)

    run(W)                                          # W untimed warm-up steps, as requested
    run(K)                                          # + one untimed pass of the schedule itself
    cal, _ = timed(1)                               # calibration pass (untimed in the result): how long is one schedule?
    repeats = int(min(max(1, math.ceil(args.min_time / max(cal, 1e-7))), 1 << 20))
    ev_s, wall_s = timed(repeats)
    timed_steps = K * repeats
    # roofline of the dominant kernel: algorithmic bytes of all launches of the timed region / its HIP-event time
    extra = getattr(env, "bench_extra_bytes", lambda n, rollout: 0)
    if mode == "rollout":
        per_pass = sum(core.bytes_per_rollout(n) + extra(n, True) for n in groups_of(K, POOL))
        launches_pass = len(groups_of(K, POOL))
    else:
        per_pass = K * (core.bytes_per_step() + extra(1, False))
        launches_pass = K
    launches = launches_pass * repeats
    bytes_total = per_pass * repeats
    n_total = core.N * world
    achieved = bytes_total / ev_s / 1e9
    kernel = getattr(env, "bench_kernel", {}).get(mode) or ("gpd_step_kernel" if mode != "rollout" else
        ("gpd_rollout1_kernel" if core.D <= 64 and core.term_obs12 is None else "gpd_rollout_kernel"))
    segments = None
    if len(marks) > 1:
        # rate of the first ~100 ms of the timed region against the rest of it (a GPU that boosts out of idle and then settles, or
        # throttles, shows here)
        head = next((i for i, (_, t) in enumerate(marks) if t >= 0.1), len(marks) - 1)
        n_h, t_h = marks[head]
        n_e, t_e = marks[-1]
        per = [((marks[i][0] - (marks[i - 1][0] if i else 0)) / max(marks[i][1] - (marks[i - 1][1] if i else 0.0), 1e-12)) for i in range(len(marks))]
        segments = {"n": len(marks), "head_ms": t_h * 1e3, "head_steps_per_s": n_h * K / t_h,
                    "rest_steps_per_s": (n_e - n_h) * K / max(t_e - t_h, 1e-12) if n_e > n_h else None,
                    "slowest_over_fastest_segment": min(per) / max(per)}
    return {
        "K": K, "W": W, "repeats": repeats, "timed_steps": timed_steps, "ev_s": ev_s, "wall_s": wall_s, "segments": segments,
        "value": n_total * core.S * timed_steps / ev_s, "value_wall": n_total * core.S * timed_steps / wall_s,
        "env_steps_per_s": n_total * timed_steps / ev_s, "us_per_step": ev_s * 1e6 / timed_steps,
        "per_gpu": [core.N * core.S * timed_steps / t for t in per_rank_s],
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "achievable": HBM_ACHIEVABLE_GBS,
                     "frac_of_achievable": achieved / HBM_ACHIEVABLE_GBS, "traffic": None, "kernel": kernel,
                     "env_steps_per_launch": K / launches_pass, "bytes_per_launch": bytes_total / launches,
                     "bytes_per_drone_per_env_step": per_pass / (core.N * K),
                     "launch_us_hip_events": ev_s * 1e6 / launches, "launches_timed": launches,
                     "clock": "HIP events on the launch stream, max over ranks"},
    }


def attach_counters(roof, key, m, core, clock_ghz):
    """Offline rocprofv3 figures (profiles/*.json) beside the live ones: FETCH_SIZE / WRITE_SIZE traffic scaled to this launch's step count, the trace's kernel average, the instruction counts behind the VALU-issue roofline."""
    issue = None
    tfile = os.path.join(REPO, "profiles", "hbm_traffic.json")
    if os.path.exists(tfile):
        rec = json.load(open(tfile)).get(key)
        if rec and rec.get("traffic_bytes"):
            prof_steps = rec.get("env_steps_per_launch", 64 if "rollout" in key else 1)
            spl = roof["env_steps_per_launch"]
            if spl == prof_steps:
                roof["traffic"] = rec["traffic_bytes"]
                roof["traffic_note"] = "per launch, rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (profiles/hbm_traffic.json)"
            elif rec.get("algorithmic_bytes"):
                # same kernel, other step count: the counters scale with the algorithmic bytes (ratio measured 1.00-1.02)
                roof["traffic"] = rec["traffic_bytes"] / rec["algorithmic_bytes"] * roof["bytes_per_launch"]
                roof["traffic_note"] = (f"counters of the {prof_steps}-step profile launch scaled by the algorithmic bytes to this "
                                        f"launch's {spl:g} steps (profiles/hbm_traffic.json)")
            roof["rocprof_kernel_avg_us"], roof["rocprof_note"] = (rec.get("rocprof_kernel_avg_ns", 0) / 1e3 if spl == prof_steps else None), rec.get("rocprof_note")
            own = rocprof_record(f"{key.split(':')[0]}:rollout{spl:g}") if spl != prof_steps else {}     # the committed trace of THIS step count (the driver's: 20)
            roof.update({k: v for k, v in own.items() if own.get("rocprof_kernel_avg_us") and v is not None})
    cfile = os.path.join(REPO, "profiles", "kernel_counters.json")
    if os.path.exists(cfile):
        rec = json.load(open(cfile)).get(key)
        if rec and rec.get("slots_per_wave_env_step"):
            waves_per_simd = math.ceil(core.N / 64 / NUM_SIMDS)
            slots, valu = rec["slots_per_wave_env_step"], rec.get("valu_per_wave_env_step")
            floor_peak = waves_per_simd * slots * 4 / (PEAK_CLOCK_GHZ * 1e3)            # us per env step at 2.4 GHz
            issue = {"bound": "valu_issue", "unit": "us/env-step", "slots_per_wave_env_step": slots,
                     "valu_per_wave_env_step": valu, "waves_per_simd": waves_per_simd, "cycles_per_slot": 4,
                     "peak_clock_ghz": PEAK_CLOCK_GHZ, "floor_us": floor_peak, "measured_us": m["us_per_step"],
                     "frac": floor_peak / m["us_per_step"], "source": "profiles/kernel_counters.json (rocprofv3 --pmc SQ_INSTS_*)"}
            if clock_ghz:
                issue["measured_clock_ghz"] = clock_ghz
                issue["frac_at_measured_clock"] = waves_per_simd * slots * 4 / (clock_ghz * 1e3) / m["us_per_step"]
            if waves_per_simd == 1:
                # ONE wave on a SIMD does not issue every 4 cycles: profiles/r01_issue_microbench.txt measures 4.5 cycles per
                # instruction with four independent chains and 5.4 for a dependent chain.  At the headline size (1024 waves on
                # 1024 SIMDs) that interval, not the 4-cycle figure, is the floor.
                ck = clock_ghz or PEAK_CLOCK_GHZ
                lo, hi = slots * 4.5 / (ck * 1e3), slots * 5.4 / (ck * 1e3)
                issue["lone_wave"] = {"cycles_per_slot": [4.5, 5.4], "clock_ghz": ck, "floor_us": [lo, hi],
                                      "frac": [lo / m["us_per_step"], hi / m["us_per_step"]],
                                      "source": "profiles/r01_issue_microbench.txt (scratch/issue.hip)"}
    hbm_floor = roof["bytes_per_launch"] / roof["env_steps_per_launch"] / (HBM_PEAK_GBS * 1e3)     # us per env step
    roof["floor_us"] = hbm_floor
    if issue is not None:
        roof["binding"] = "valu_issue" if issue["floor_us"] > hbm_floor else "hbm"
    return issue


def dropin_single_env(device, steps=2420):
    """BASELINE config 1's shape through the DROP-IN class: `HoverAviary()` with its defaults (30 Hz control / 240 Hz physics,
    Physics.PYB = explicit integrator + ground plane here), ActionType.ONE_D_RPM, a ~ U(-1, 1) of shape (1, 1), 2 420 `step()` calls
    with a reset at every episode end (examples/learn.py:54-58; envs/BaseAviary.py:509-519 is the per-step read-back the reference
    pays).  Wall clock per call -- this path is latency, not bandwidth: the aviary's state lives in host-visible memory, a step is
    one gpd_step_sync call (the launch + a spin on the word the kernel writes when it is done) + numpy reads -- and where it goes."""
    import warnings
    from gym_pybullet_drones_amd.envs import HoverAviary
    from gym_pybullet_drones_amd.utils.enums import ActionType
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        env = HoverAviary(act=ActionType.ONE_D_RPM, device=device)
    acts = np.random.default_rng(0).uniform(-1, 1, size=(steps, 1, 1)).astype(np.float32)
    env.reset(seed=0)
    for k in range(64):
        env.step(acts[k])
    env.reset(seed=0)
    torch.cuda.synchronize()
    episodes, t0 = 0, time.perf_counter()
    for k in range(steps):
        _, _, term, trunc, _ = env.step(acts[k])
        if term or trunc:
            env.reset()
            episodes += 1
    dt = time.perf_counter() - t0
    core = env._core

    def per_call_us(fn, n=1000):
        t = time.perf_counter()
        for _ in range(n):
            fn()
        return (time.perf_counter() - t) / n * 1e6
    parts = {"gpd_step_sync_launch_and_wait": per_call_us(core.step_host), "numpy_kinematic_refresh": per_call_us(env._updateAndStoreKinematicInformation),
             "observation_row_with_history": per_call_us(env._computeObs)}
    S = int(env.PYB_STEPS_PER_CTRL)
    out = {"us_per_step": dt / steps * 1e6, "env_steps_per_s": steps / dt, "value": steps * S / dt, "unit": "drone-steps/s", "steps": steps,
           "episodes_ended": episodes, "state_memory": "host-visible (page-locked, device-mapped)" if core.host_visible else "HBM",
           "breakdown_us": parts,
           "what": "for k in range(2420): HoverAviary(act=ONE_D_RPM).step(a[k]) (+ reset at episode ends), wall clock; one gpd_step_sync call "
                   "(launch + wait on the kernel's completion word) per step, no device-to-host copy, no second launch"}
    env.close()
    try:        # the reference's OWN Python on the same schedule (quoted: /root/reference does not exist on this box)
        rec = json.load(open(os.path.join(REPO, "profiles", "r05_reference_python_dyn_cpu.json")))
        out["reference_python_us_per_step"] = 1e6 * S / rec["value"]
        out["speedup_over_reference_python"] = out["value"] / rec["value"]
        out["reference_python_note"] = ("the reference's unmodified HoverAviary(physics=Physics.DYN) over oracle/pybullet_shim.py, same schedule, "
                                        f"1 core of {rec.get('host_cpu')}, timed in the build container (profiles/r05_reference_python_dyn_cpu.json)")
    except Exception as e:      # noqa: BLE001
        out["reference_python_note"] = f"no committed figure ({type(e).__name__})"
    return out


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4096)
    ap.add_argument("--warmup", type=int, default=256)
    ap.add_argument("--workload", default="hover65536_240hz", choices=sorted(WORKLOADS))
    ap.add_argument("--mode", default="rollout", choices=["rollout", "graph", "eager"], help="see the module docstring")
    ap.add_argument("--min-time", type=float, default=0.25, help="repeat the K-step schedule until the timed region lasts this long [s]")
    ap.add_argument("--no-parity", action="store_true", help="skip the replay of one schedule through the float64 C oracle")
    ap.add_argument("--allgather", action="store_true", help="all-gather the obs shards over RCCL")
    ap.add_argument("--allgather-impl", default="auto", choices=["auto", "native", "torch"], help="native: gpd_allgather_obs (ncclAllGather); torch: torch.distributed")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-second-leg", action="store_true", help="skip the one-launch-per-step and Python-loop legs beside the rollout headline")
    ap.add_argument("--no-dropin-leg", action="store_true", help="skip `dropin_single_env` (HoverAviary().step(), BASELINE config 1's shape)")
    ap.add_argument("--scale-suite", action="store_true", help="also run BASELINE configs 4 and 5 in the same job (`suite`; default with --gpus > 1)")
    ap.add_argument("--no-suite", action="store_true")
    ap.add_argument("--suite-timeout", type=float, default=300.0, help="seconds the suite may take before the headline line is printed without it")
    ap.add_argument("--hbm-leg-time", type=float, default=2.2, help="seconds of timed region of the `hbm_saturating` leg (its whole launches: >= 2 s)")
    ap.add_argument("--hbm-leg-reallocations", type=int, default=10, help="re-runs of the `hbm_saturating` leg on freshly allocated buffers (0.15 s each)")
    ap.add_argument("--placement-search", action="store_true", help="rollout mode: the launch's blocks at offsets chosen by probing (placement.py; the hbm leg does)")
    ap.add_argument("--placement-target", type=float, default=0.755, help="the search stops at the first layout that reaches this fraction of 8 TB/s")
    ap.add_argument("--segment-events", type=int, default=0, help="extra events inside the timed region: first-100-ms vs steady-state rates (`segments`)")
    ap.add_argument("--no-hbm-leg", action="store_true", help="skip `hbm_saturating` (hover4m_240hz: a working set the 256 MiB Infinity Cache cannot hold)")
    ap.add_argument("--dry-run-topology", action="store_true", help="only bring the job up (process group, RCCL, one 12-float all-gather), print what the ranks see, exit")
    ap.add_argument("--init-timeout", type=float, default=180.0, help="seconds the process group / RCCL bring-up may take before the job gives up with a readable line")
    args = ap.parse_args(argv)
    if args.steps < 1 or args.warmup < 0:
        ap.error("--steps must be >= 1 and --warmup >= 0")
    if args.gpus < 1:
        ap.error("--gpus must be >= 1")
    return args


def free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(args, argv):
    """`python bench.py --gpus N` with N > 1 and no torchrun around it: start the N ranks ourselves (one process per GPU,
    `torch.distributed.run` on 127.0.0.1 and a free port) with the very same arguments, pass rank 0's ONE JSON line through, and
    exit with the job's return code.  A node with fewer than N devices is an error, not a silent smaller run (the
    GPD_BENCH_SINGLE_DEVICE test hook puts every rank on device 0).  Before the real job the same ranks are brought up once with
    `--dry-run-topology` (seconds): a node whose RCCL cannot form the communicator says so before anything is timed."""
    import subprocess
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if have < args.gpus and not os.environ.get("GPD_BENCH_SINGLE_DEVICE"):
        print(f"[bench] --gpus {args.gpus} but this node shows {have} device(s): refusing to report a smaller run as {args.gpus} GPUs",
              file=sys.stderr)
        raise SystemExit(2)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", GPD_BENCH_SELF_LAUNCHED="1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, host_threads() // args.gpus)))

    def cmd(extra):
        return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
                "--master-port", str(free_port()), os.path.abspath(__file__)] + list(argv) + extra
    if not args.dry_run_topology and not os.environ.get("GPD_BENCH_NO_TOPOLOGY_CHECK"):
        dry = subprocess.run(cmd(["--dry-run-topology"]), env=env, capture_output=True, text=True)
        line = next((l for l in (dry.stdout or "").splitlines() if l.startswith("{")), None)
        print(f"[bench] topology dry run: rc {dry.returncode} {line or (dry.stderr or '')[-2000:]}", file=sys.stderr)
        if dry.returncode != 0:
            if line:
                print(line)
            raise SystemExit(dry.returncode)
    c = cmd([])
    print(f"[bench] launching {args.gpus} ranks: {' '.join(c)}", file=sys.stderr)
    sys.stderr.flush()
    raise SystemExit(subprocess.run(c, env=env).returncode)


class Job:
    """What every workload of one bench.py process shares: the process group, this rank's device, and what the ranks say about the
    topology they found (`gym_pybullet_drones_amd.dist.bring_up`; printed into the line, and all there is to a `--dry-run-topology`).
    A bring-up that fails or hangs ends in ONE readable line from rank 0 and exit code 3 everywhere, not in the driver's timeout."""
    def __init__(self, args):
        from gym_pybullet_drones_amd import dist as gdist
        # (GPD_DIST_BACKEND / GPD_BENCH_SINGLE_DEVICE: test hooks -- run the multi-rank code path with gloo on one GPU)
        self.backend = os.environ.get("GPD_DIST_BACKEND", "nccl")
        world_env, rank_env = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
        assert torch.cuda.is_available(), "bench.py needs a GPU (the product path has no CPU fallback)"
        local = int(os.environ.get("LOCAL_RANK", "0"))
        # (LOCAL_RANK modulo the visible devices: a launcher that gives every rank ONE visible device means index 0 on each)
        self.device = torch.device("cuda", local % max(torch.cuda.device_count(), 1) if world_env > 1 and not os.environ.get("GPD_BENCH_SINGLE_DEVICE") else 0)
        torch.cuda.set_device(self.device)
        self.topology = {}

        def failed(what):
            return {"metric": BASELINE_METRIC, "value": None, "n_gpus": world_env, "error": what, "rccl_warnings": gdist.rccl_debug_tail(),
                    "topology": self.topology}
        dog = Watchdog(args.init_timeout, type("R", (), {"rank": rank_env})(), {},
                       lambda o: o.update(failed(f"process group / RCCL bring-up not finished after {args.init_timeout:.0f} s")), code=3) if world_env > 1 else None
        try:
            gdist.bring_up(self.backend, self.device, self.topology)
        except Exception as e:          # noqa: BLE001
            if dog is not None:
                dog.done()
            if rank_env == 0:
                print(json.dumps(failed(f"bring-up failed: {type(e).__name__}: {e}"[:600])))
                sys.stdout.flush()
            os._exit(3)
        if dog is not None:
            dog.done()
        t = self.topology
        self.rank, self.world, self.local = t["rank"], t["world_size"], t["local_rank"]
        self.ranks_in_group, self.native_ranks_seen, self.native_note = t["ranks_in_process_group"], t["n_ranks_seen_by_rccl"], t["native_comm_note"]
        if args.gpus != self.world:
            # (reached only when a launcher set WORLD_SIZE to something else than --gpus: the line reports what really ran)
            if self.rank == 0:
                print(f"[bench] --gpus {args.gpus} but WORLD_SIZE={self.world}: reporting {self.world}", file=sys.stderr)
            args.gpus = self.world

    def dry_run(self):
        """`--dry-run-topology`: one 12-float all-gather, the topology block as ONE JSON line, exit code 0 / 3."""
        from gym_pybullet_drones_amd import dist as gdist
        ok, note = gdist.dry_run_exchange(self.topology, self.backend, self.device)
        if self.rank == 0:
            print(json.dumps({"dry_run_topology": True, "ok": ok, "n_gpus": self.world, "allgather_12_floats": note, "topology": self.topology}))
            sys.stdout.flush()
        return 0 if ok else 3


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args, argv)
    job = Job(args)
    if args.dry_run_topology:
        rc = job.dry_run()
        if job.world > 1:
            torch.distributed.barrier()
            torch.distributed.destroy_process_group()
        raise SystemExit(rc)
    suite = (args.scale_suite or (job.world > 1 and args.workload == "hover65536_240hz")) and not args.no_suite
    out = run_workload(args, job)
    if args.workload == "hover65536_240hz" and not args.no_hbm_leg:
        hbm_leg(args, job, out)
    if suite:
        run_suite(args, job, out)
    if job.rank == 0:
        print(json.dumps(out))
        sys.stdout.flush()
    if job.world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


class Watchdog:
    """Bounds a part of the job that may hang (the bring-up, the HBM-saturating leg, the multi-GPU suite): when `seconds` pass before
    `done()`, rank 0 prints the line it already holds -- `note(out)` has written what happened into it -- and every rank leaves the
    process.  A rank that raised before a collective leaves the others waiting in it; that must not cost the headline."""

    def __init__(self, seconds, job, out, note, code=0):
        import threading
        self._done = threading.Event()

        def watch():
            if self._done.wait(seconds):
                return
            if job.rank == 0:
                note(out)
                print(json.dumps(out), flush=True)
            os._exit(code)

        threading.Thread(target=watch, daemon=True).start()

    def done(self):
        self._done.set()


def hbm_leg(args, job, out):
    """The headline's working set (99 MB per 20-step launch, re-used by every launch) fits the 256 MiB Infinity Cache, and the
    FETCH_SIZE / WRITE_SIZE counters count L2 -> fabric requests whether HBM or the cache serves them: "of the HBM roofline"
    next to it is a fabric-side rate.  This leg puts a number beside it that the cache cannot serve: the same kernel, the same
    schedule, 4 194 304 drones per GPU (13.9 GB of observation rows per 64-step launch, 218 MB of state)."""
    a = argparse.Namespace(**vars(args))
    a.workload, a.no_cpu_baseline, a.no_second_leg, a.no_dropin_leg, a.allgather = "hover4m_240hz", True, True, True, False
    a.placement_search = not os.environ.get("GPD_BENCH_NO_PLACEMENT_SEARCH")       # (the A/B: the driver's allocation as it comes)
    # >= 2 s of timed region, seen in 48 pieces (0.1 s = 32 launches cannot tell a boost clock from a steady one); --hbm-leg-time shortens it for tests
    a.min_time = float(getattr(args, "hbm_leg_time", 2.2))
    a.segment_events = 48
    # 64 steps per launch whatever the headline's --steps: the block is about the rate HBM serves this kernel at, and at 4M drones a
    # 20-step launch (16 rounds of resident workgroups, each starting and ending together) reads 0.63-0.71 where the 64-step one reads
    # 0.72-0.75 (profiles/r04_hbm_leg_steps_per_launch.txt)
    a.steps, a.warmup = min(256, max(64, args.steps // 64 * 64)), 64
    # (the parity replay copies every replayed step's rows to the host, twice as float64: 8 steps of 4M drones are 1.6 + 3.2 GB)
    a.parity_max_steps = 8
    limit = min(120.0, args.suite_timeout)
    dog = Watchdog(limit, job, out, lambda o: o.__setitem__("hbm_saturating", {"error": f"not finished after {limit:.0f} s: line printed without it"}))
    try:
        r = run_workload(a, job)
    except Exception as e:          # noqa: BLE001 -- reported; the headline survives
        r = {"error": f"{type(e).__name__}: {e}"[:300]}
    finally:
        dog.done()
    # The extras have their own guard and their own budget: a failure or a time-out in them leaves `on_fresh_allocations` partial and
    # the leg's figure -- already measured -- in the line.  Every rank re-runs on its own (no collective inside): the same
    # environment, its rollout blocks freed and placed again -- a new arena searched (or, without the search, new plain allocations).
    copy, again, again_note = None, [], None
    if (r is None or "error" not in r) and getattr(job, "last_env", None) is not None:
        n_again, budget = int(getattr(args, "hbm_leg_reallocations", 10)), min(90.0, args.suite_timeout)
        dog2 = Watchdog(budget, job, out, lambda o: o.__setitem__("hbm_saturating", hbm_block(r, copy, again, f"the re-runs on fresh allocations were cut after {budget:.0f} s")))
        try:
            import gc
            if job.rank == 0:
                copy = copy_probe(job.device)
            env, core = job.last_env, job.last_env.core
            for _ in range(n_again):
                env.placement_arena = acts = None
                core.__dict__.get("_rollout_cache", {}).clear()
                gc.collect()
                torch.cuda.empty_cache()
                rep = None
                if a.placement_search:
                    from gym_pybullet_drones_amd.placement import place_rollout
                    env.placement_arena, rep = place_rollout(core, POOL, target=args.placement_target, accept=min(args.placement_target, 0.75))
                    acts = env.placement_arena.actions
                else:
                    acts = torch.rand((POOL, core.N, core.A), device=job.device) * 2 - 1
                core.rollout(acts, update_latest=False)
                n = 45                                                      # (0.15 s of 64-step launches)
                sec = event_seconds(lambda: core.rollout(acts, update_latest=False), n)
                again.append({"frac": core.bytes_per_rollout(POOL) / sec / (HBM_PEAK_GBS * 1e9), "launch_us_hip_events": sec * 1e6, "launches": n,
                              **({"probes": rep["probes"], "arenas_tried": rep["arenas_tried"], "levels_seen": rep["seen"]} if rep else {})})
        except Exception as e:      # noqa: BLE001
            again_note = f"stopped after {len(again)} re-runs: {type(e).__name__}: {e}"[:300]
        finally:
            dog2.done()
            job.last_env = None
    if job.rank == 0:
        out["hbm_saturating"] = r if "error" in r else hbm_block(r, copy, again, again_note)
        if "error" not in r:
            out["roofline"]["traffic_scope"] = ("the headline's working set (%.0f MB per launch, re-used by every launch) is Infinity-Cache "
                                                "resident: achieved / traffic are fabric-side rates; see hbm_saturating for the HBM-served figure"
                                                % (out["roofline"]["bytes_per_launch"] / 1e6))


def hbm_block(r, copy, again, again_note):
    roof = r["roofline"]
    fr = [roof["frac"]] + [x["frac"] for x in again]
    return {
        "workload": "hover4m_240hz", "drones_per_gpu": r["config"]["envs_per_gpu"], "value": r["value"], "unit": r["unit"], "ms_per_step": r["ms_per_step"],
        "timed_steps": r["timed_steps"], "timed_region_ms": r["timed_region_ms"], "kernel": roof["kernel"],
        "env_steps_per_launch": roof["env_steps_per_launch"], "launch_us_hip_events": roof["launch_us_hip_events"],
        "bytes_per_launch": roof["bytes_per_launch"], "achieved": roof["achieved"], "peak": roof["peak"], "unit_bw": "GB/s",
        "frac": roof["frac"], "frac_of_achievable": roof["frac_of_achievable"], "traffic": roof.get("traffic"),
        "segments": r.get("segments"), "clock_ghz_after": r.get("clock_ghz_after_timed_region"),
        # how the blocks of the launch were placed: by search inside one arena, the launch itself as the probe (or: as the allocator handed them out)
        "placement": r.get("placement") or {"what": "as allocated (no search)"},
        # the same environment with its blocks freed and placed again (a fresh arena searched / fresh plain allocations), 0.15 s each:
        # does the rate depend on the allocation?
        "on_fresh_allocations": again, "frac_range_over_allocations": [min(fr), max(fr)], "frac_spread_over_allocations": (max(fr) - min(fr)) / max(fr),
        "allocations_seen": len(fr), **({"on_fresh_allocations_note": again_note} if again_note else {}),
        # what a plain device-to-device copy reaches on THIS box in THIS process ("measure achievable ... and quote both")
        "copy_probe": copy, "frac_of_measured_copy": (roof["achieved"] / copy["gbs"]) if copy and copy.get("gbs") else None,
        **rocprof_record("hover4m_240hz:rollout64"),
        "parity": {k: r["parity"].get(k) for k in ("checked_steps", "max", "tolerance", "ok", "flag_mismatch_frac", "error") if k in r.get("parity", {})},
        "note": "working set of one launch >> the 256 MiB Infinity Cache: this rate is served by HBM"}


def event_seconds(fn, reps):
    """seconds per call of `fn`, HIP events on the current stream around `reps` back-to-back calls"""
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    ev0.record()
    for _ in range(reps):
        fn()
    ev1.record()
    torch.cuda.synchronize()
    return ev0.elapsed_time(ev1) * 1e-3 / reps


def copy_probe(device, mib=1024, reps=20):
    """Device-to-device copy bandwidth (read + write bytes / time) of a buffer the Infinity Cache cannot hold, and a write-only fill of
    the same buffer (the rollout kernel's traffic is 76 % stores): the streaming rates this box reaches right now, HIP events."""
    try:
        n = mib * (1 << 20) // 4
        a = torch.empty(n, dtype=torch.float32, device=device).normal_()
        b = torch.empty_like(a)
        res = {}
        for name, fn, nbytes in (("gbs", lambda: b.copy_(a), 2 * n * 4), ("fill_gbs", lambda: b.fill_(1.0), n * 4)):
            event_seconds(fn, 3)
            res[name] = nbytes / event_seconds(fn, reps) / 1e9
        return dict(res, mib=mib, reps=reps, what="torch Tensor.copy_ device to device (read + write bytes); fill_gbs: Tensor.fill_ of the same buffer (write only)")
    except Exception as e:          # noqa: BLE001
        return {"error": f"{type(e).__name__}: {e}"[:160]}


def rocprof_record(key):
    """The committed rocprofv3 figures for one kernel (profiles/*hbm_reconcile.json, written on a GPU box by one script on one lease):
    quoted in the bench line so that the line and the profile can be held against each other."""
    for name in ("r06_hbm_reconcile.json", "r05_hbm_reconcile.json"):
        try:
            rec = json.load(open(os.path.join(REPO, "profiles", name)))["keys"][key]
            return {"rocprof_kernel_avg_us": rec["rocprof_kernel_avg_us"], "rocprof_frac": rec.get("rocprof_frac"), "rocprof_note": rec.get("note"), "rocprof_file": "profiles/" + name}
        except Exception:           # noqa: BLE001
            continue
    return {"rocprof_kernel_avg_us": None, "rocprof_note": "no committed reconciliation"}


#: the multi-GPU configurations of BASELINE.json (configs 4 and 5), per GPU
SUITE = ("hover65536x8_allgather", "multihover2x16384x8")


def run_suite(args, job, out):
    """The other multi-GPU lines in the same job, compact, under out["suite"].  A watchdog bounds the whole suite: when it
    fires, rank 0 prints the headline line it already holds (the suite entry says what happened) and every rank leaves -- a
    hang in a collective of a workload that has never met this node must not cost the headline."""
    from gym_pybullet_drones_amd import dist as gdist
    results = {}
    if job.rank == 0:
        out["suite"] = results
    dog = Watchdog(args.suite_timeout, job, out,
                   lambda o: results.__setitem__("error", f"suite not finished after {args.suite_timeout:.0f} s: line printed without the rest"))
    try:
        for name in SUITE:
            a = argparse.Namespace(**vars(args))
            a.workload, a.no_cpu_baseline, a.no_hbm_leg, a.no_dropin_leg, a.allgather = name, True, True, True, False
            t0 = time.perf_counter()
            try:
                r = run_workload(a, job)
            except Exception as e:          # noqa: BLE001 -- reported; the headline survives
                r = {"error": f"{type(e).__name__}: {e}"[:300], "rccl_warnings": gdist.rccl_debug_tail()}
            ok = gdist.all_ranks_ok(r is None or "error" not in r, device=job.device)
            if job.rank == 0:
                if "error" in r or not ok:
                    results[name] = {"error": r.get("error", "failed on another rank"), "rccl_warnings": r.get("rccl_warnings")}
                else:
                    keep = ("metric", "value", "unit", "n_gpus", "steps", "ms_per_step", "scaling", "per_gpu", "timed_steps", "value_wall")
                    c = {k: r[k] for k in keep if k in r}
                    c["roofline"] = {k: r["roofline"].get(k) for k in ("bound", "achieved", "peak", "unit", "frac", "kernel", "env_steps_per_launch")}
                    c["config"] = {k: r["config"].get(k) for k in ("workload", "total_drones", "obs_allgather", "allgather_impl", "allgather_note",
                                                                   "n_ranks_seen_by_rccl", "ranks_in_process_group", "launch", "mode")}
                    for k in ("one_launch_per_step", "parity", "without_allgather"):
                        if k in r:
                            c[k] = r[k] if k != "one_launch_per_step" else {q: r[k][q] for q in ("value", "us_per_step", "launch")}
                    c["wall_s"] = time.perf_counter() - t0
                    results[name] = c
            if not ok:
                break
    finally:
        dog.done()


def rehearsal_scale(w):
    """GPD_BENCH_E_DIV=k (a REHEARSAL hook, e.g. eight ranks sharing one device in tests/test_gpu_multirank.py): every workload runs
    with 1/k of its aviaries (or, one world: of its drones).  The line then says so (`config.rehearsal_divisor`) and its `value`
    is not a measurement of the named configuration."""
    k = int(os.environ.get("GPD_BENCH_E_DIV", "1") or 1)
    if k <= 1:
        return w, 1
    return dict(w, **({"D": max(4096, w["D"] // k)} if w.get("swarm") else {"E": max(256, w["E"] // k)})), k


def run_workload(args, job):
    """Measure ONE workload on every rank of the job; rank 0 gets the JSON object, the others None."""
    from gym_pybullet_drones_amd import dist as gdist
    backend, rank, world, device = job.backend, job.rank, job.world, job.device
    w, rehearsal_div = rehearsal_scale(WORKLOADS[args.workload])
    want_gather = bool(args.allgather or w.get("allgather"))
    mode = w["pick_mode"](args.mode, world, backend) if w.get("pick_mode") else args.mode
    env = make_env(w, device, seed=1000 + rank * 16, world=world, rank=rank, job=job)
    actions = make_actions(w, env, device, seed=2000 + rank * 16, pool=POOL)
    core = env.core
    placement = None
    if getattr(args, "placement_search", False) and mode == "rollout":
        # where the launch's blocks sit in HBM decides its rate (profiles/r06_hbm_placement_cause.md): the library carves them out of
        # one arena at offsets it chooses by probing with the launch itself (gym_pybullet_drones_amd/placement.py), before anything is timed
        from gym_pybullet_drones_amd.placement import place_rollout
        arena, placement = place_rollout(core, POOL, target=args.placement_target, accept=min(args.placement_target, 0.75))
        arena.actions.copy_(actions.view_as(arena.actions))
        actions = arena.actions.view(actions.shape)
        env.placement_arena = arena                     # (owns the blocks: lives as long as the environment)
        env.reset()
    gather, impl, gather_note, ranks_seen = None, None, None, None
    if want_gather:
        impl = args.allgather_impl
        if impl == "auto":
            impl = "native" if (world == 1 or backend == "nccl") else "torch"
        if impl == "native":
            # ONE communicator for every count (NativeComm.shared), exercised once before anything is timed; a failure on
            # ANY rank sends ALL ranks to torch.distributed's all-gather together instead of aborting the job
            err = None
            try:
                gather = gdist.NativeObsAllGather(core.N, 12, device=device)
                gather(core.obs12)
                torch.cuda.synchronize()
                ranks_seen = gather.nc.ranks_seen
            except Exception as e:      # noqa: BLE001 -- reported in the JSON line
                err = f"{type(e).__name__}: {e}"[:300]
            if not gdist.all_ranks_ok(err is None, device=device):
                gather_note = f"native RCCL all-gather unavailable ({err or 'failed on another rank'}): fell back to torch.distributed"
                impl, gather = "torch", None
        if impl == "torch":
            gather = gdist.ObsAllGather(core.N, 12, device=device)

    second = eager = plain = None
    if mode == "rollout" and not args.no_second_leg:
        # (a gather staged through host memory -- the gloo test hook -- cannot be captured in a hipGraph: that leg then runs without it)
        second = measure("graph", args, env, actions, None if getattr(gather, "_stage", False) else gather, device, world)
        env.reset()
        if world == 1 and not w.get("builder"):
            # ... and the plain Python loop `for ...: env.step(action)`, one host launch per step, nothing captured: what a user's own
            # loop gets (host-bound: the interpreter and the ctypes call take longer than the kernel)
            a2 = argparse.Namespace(**vars(args))
            a2.min_time, a2.segment_events = min(args.min_time, 0.1), 0
            eager = measure("eager", a2, env, actions, None, device, world)
            env.reset()
    if gather is not None:
        # BASELINE config 4 asks for both: first the same schedule WITHOUT the collective
        plain = measure(mode, args, env, actions, None, device, world)
        env.reset()
    m = measure(mode, args, env, actions, gather, device, world)
    job.last_env = env                      # (the hbm leg re-places this environment's blocks for its re-runs)

    parity = None
    if w.get("parity"):                     # (bench_extra.py's workloads bring their own checker; collective where the world is shared)
        parity = w["parity"](args, job, env)
    elif rank == 0 and not args.no_parity and mode == "rollout":
        try:
            from oracle.bench_checks import parity_check       # checker code: the product path stays oracle-free
            parity = parity_check(w, env, actions, args.steps, POOL, launch_rollout, max_steps=getattr(args, "parity_max_steps", 256))
        except Exception as e:          # noqa: BLE001 -- the checker must never take the measurement down with it
            parity = {"error": f"{type(e).__name__}: {e}"[:300]}

    clock_ghz = None
    try:        # the shader clock a one-wave-per-SIMD FMA chain runs at right after the timed region (diagnostics)
        ghz, nsf = ctypes.c_double(), ctypes.c_double()
        if core.lib.gpd_clock_probe(ctypes.byref(ghz), ctypes.byref(nsf), ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)) == 0:
            clock_ghz = ghz.value
    except Exception:       # noqa: BLE001
        pass
    dropin = None
    if rank == 0 and world == 1 and args.workload == "hover65536_240hz" and not args.no_dropin_leg:
        try:
            dropin = dropin_single_env(device)
        except Exception as e:          # noqa: BLE001
            dropin = {"error": f"{type(e).__name__}: {e}"[:300]}
    if rank != 0:
        if w.get("finish"):
            w["finish"](None, args, job, env, m, clock_ghz)
        return None

    spl = m["roofline"]["env_steps_per_launch"]
    launch = {"rollout": f"rollout{int(spl) if spl == int(spl) else spl:g}", "graph": "graph", "eager": "eager"}[mode]
    metric = BASELINE_METRIC if args.workload == "hover65536_240hz" else \
        (f"env steps/sec (whole node), {args.workload}: {w['E']} aviaries x {w['D']} drone(s) per GPU, {w['ctrl']} Hz control / 240 Hz physics")
    out = {
        "metric": metric,
        "value": m["value"], "unit": "drone-steps/s", "n_gpus": world, "steps": m["K"], "warmup": m["W"],
        "ms_per_step": m["ev_s"] * 1e3 / m["timed_steps"], "higher_is_better": True,
        "scaling": w.get("scaling", "weak"), "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "repeats": m["repeats"], "timed_steps": m["timed_steps"], "timed_region_ms": m["ev_s"] * 1e3,
        "wall_ms_per_step": m["wall_s"] * 1e3 / m["timed_steps"], "value_wall": m["value_wall"],
        "config": {"workload": args.workload, "envs_per_gpu": w["E"], "drones_per_env": core.D,
                   **({"rehearsal_divisor": rehearsal_div, "rehearsal_note": "GPD_BENCH_E_DIV: NOT the named configuration's size"} if rehearsal_div > 1 else {}),
                   "total_drones": core.N * world,
                   "physics": "DYN" + "".join(n for b, n in ((1, "+GND"), (2, "+DRAG"), (4, "+DW"), (8, "+GROUND_PLANE"), (16, "+BULLET_DAMPING")) if core.physics_flags & b),
                   "physics_flags": core.physics_flags, "pyb_freq": 240, "ctrl_freq": w["ctrl"], "substeps_per_step": core.S, "action": w["act"],
                   "task": w["task"], "auto_reset": True, "mode": mode, "launch": launch, "full_obs": w.get("full_obs", False),
                   "obs_allgather": want_gather, "allgather_impl": impl, "allgather_note": gather_note,
                   "n_ranks_seen_by_rccl": ranks_seen if ranks_seen is not None else job.native_ranks_seen,
                   "ranks_in_process_group": job.ranks_in_group, "process_group_backend": backend if world > 1 else None,
                   **({"native_comm_note": job.native_note} if job.native_note else {}),
                   "launcher": "bench.py self-launch (torch.distributed.run)" if os.environ.get("GPD_BENCH_SELF_LAUNCHED") else
                               ("torch.distributed.run" if world > 1 else "single process"),
                   "topology": job.topology, "env_steps_per_s": m["env_steps_per_s"]},
        "roofline": m["roofline"],
        "per_gpu": {"unit": "drone-steps/s", "values": m["per_gpu"], "min": min(m["per_gpu"]), "max": max(m["per_gpu"]),
                    "note": "every rank's own units / its own HIP-event time of the same timed region; `value` uses the slowest rank's time"},
    }
    if placement is not None:
        out["placement"] = {k: v for k, v in placement.items() if k != "all_probes"}
    if plain is not None:
        out["without_allgather"] = {"value": plain["value"], "unit": "drone-steps/s", "us_per_step": plain["us_per_step"],
                                    "per_gpu": plain["per_gpu"], "frac": plain["roofline"]["frac"]}
        per_step = 12 * 4 * core.N
        out["allgather"] = {"bytes_per_rank_per_env_step": per_step, "bytes_per_collective": per_step * spl if mode == "rollout" else per_step,
                            "us_per_step_added": m["us_per_step"] - plain["us_per_step"]}
    issue = attach_counters(out["roofline"], f"{args.workload}:{'rollout64' if mode == 'rollout' else mode}", m, core, clock_ghz)
    if issue is not None:
        out["roofline_valu_issue"] = issue
    if clock_ghz:
        out["shader_clock_ghz_probe"] = out["clock_ghz_after_timed_region"] = clock_ghz
    if m.get("segments"):
        out["segments"] = m["segments"]
    if second is not None:
        sec = {"value": second["value"], "unit": "drone-steps/s", "steps": second["K"], "repeats": second["repeats"],
               "timed_steps": second["timed_steps"], "us_per_step": second["us_per_step"],
               "launch": f"graph (hipGraph of {min(second['K'], POOL)} single-step launches)", "roofline": second["roofline"]}
        si = attach_counters(sec["roofline"], f"{args.workload}:graph", second, core, clock_ghz)
        if si is not None:
            sec["roofline_valu_issue"] = si
        out["one_launch_per_step"] = sec
    if eager is not None:
        out["python_step_loop"] = {"value": eager["value_wall"], "unit": "drone-steps/s", "us_per_step": eager["wall_s"] * 1e6 / eager["timed_steps"],
                                   "us_per_step_hip_events": eager["us_per_step"], "timed_steps": eager["timed_steps"],
                                   "what": "for i in range(n): env.step(action[i]) -- one gpd_step call per step from Python, no hipGraph; wall clock around the loop + one synchronize"}
    if dropin is not None:
        out["dropin_single_env"] = dropin
    if parity is not None:
        out["parity"] = parity
    if w.get("finish"):                 # (bench_extra.py: the one-world config block, its roofline, its CPU baseline)
        w["finish"](out, args, job, env, m, clock_ghz)
    elif not args.no_cpu_baseline and world == 1:
        from oracle.bench_checks import cpu_baseline           # the oracle timed on this box's host cores (reported, not the target)
        out["cpu_baseline"] = cpu_baseline(w, phys=core.physics_flags)
    return out


if __name__ == "__main__":
    main()
