#!/usr/bin/env python3
"""gpd_rollout1_kernel compiled for ONE aviary size (template parameter DC = 2, 8; one sub-step per step): the size tests, the mates
loops of the downwash and the task sums and the sub-step loop fold away (step_rollout.hip).  (1) bit for bit the library before the
change (scratch/exp_r06/libgpd_before_unroll.so, built from fc51dc0); (2) time per launch, interleaved.
(A first version -- `if (D == 8)` straight-line blocks inside env_step, every kernel paying for the extra code -- made stacks of 8
6 % faster and pairs 6 % SLOWER, aviaries of 12 +4.7 %: gpurun_out/r06u/ab_unroll.log of that run, profiles/r06_ab_sized_rollout.json.)"""
import hashlib
import json
import os
import subprocess
import sys

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
LIBS = {"before": "scratch/exp_r06/libgpd_before_unroll.so", "pairs": "gym_pybullet_drones_amd/csrc/libgpd.so"}
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, R)
    import numpy as np
    import torch
    import bench
    from gym_pybullet_drones_amd.envs import VectorAviary
    from gym_pybullet_drones_amd.utils.enums import ActionType
    out = {}
    dev = torch.device("cuda", 0)
    for D, phys in ((2, 4), (2, 7), (2, 1), (8, 4), (8, 7), (4, 7), (16, 7), (64, 7), (3, 7), (5, 7), (12, 7), (100, 7), (256, 7)):
        for act, S in (("rpm", 1), ("pid", 1), ("rpm", 2), ("vel", 8)):
            E = max(4096 // D, 2)
            rng = np.random.default_rng(D)
            xyz, rpy = bench.stack_scene(rng, E, D)
            env = VectorAviary(E, D, initial_xyzs=xyz, initial_rpys=rpy, physics=phys, pyb_freq=240, ctrl_freq=240 // S, act=ActionType(act),
                               task="multihover", auto_reset=True, track_rpm=True, device=dev)
            g = torch.Generator(device=dev); g.manual_seed(5)
            a = torch.rand((64, E, D, env.ACT_DIM), generator=g, device=dev) * 2 - 1
            if act == "pid":
                a = a * 0.5; a[..., 2] += 1.0
            if act == "vel":
                a[..., 3] = a[..., 3].abs()
            h = hashlib.sha256()
            o, r, te, tr = env.core.rollout(a.contiguous(), update_latest=False)
            for t in (o, r, te, tr, env.core.kin_store):
                h.update(t.cpu().numpy().tobytes())
            for k in range(8):
                o, r, te, tr = env.core.step(a[k].contiguous())
                for t in (o, r, te, tr):
                    h.update(t.cpu().numpy().tobytes())
            out[f"D{D}_phys{phys}_{act}_S{S}"] = h.hexdigest()[:16]
    # time per env step of a 64-step rollout, 65 536 drones in aviaries of D (HIP events, 30 launches)
    for D in (2, 3, 4, 8, 12, 16, 100):
        E = 65536 // D
        rng = np.random.default_rng(D)
        xyz, rpy = bench.stack_scene(rng, E, D)
        env = VectorAviary(E, D, initial_xyzs=xyz, initial_rpys=rpy, physics=7, pyb_freq=240, ctrl_freq=240, act=ActionType("rpm"),
                           task="multihover", auto_reset=True, track_rpm=True, device=dev)
        a = (torch.rand((64, E, D, 4), device=dev) * 2 - 1).contiguous()
        for _ in range(3):
            env.core.rollout(a, update_latest=False)
        out[f"us_per_step_D{D}"] = bench.event_seconds(lambda: env.core.rollout(a, update_latest=False), 30) * 1e6 / 64
    print(json.dumps(out))
    raise SystemExit(0)

digests = {}
for v, lib in LIBS.items():
    p = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=dict(os.environ, GPD_LIB=os.path.join(R, lib)), capture_output=True, text=True)
    line = next((l for l in p.stdout.splitlines() if l.startswith("{")), None)
    if line is None:
        print(v, "FAILED", p.stderr[-800:])
        raise SystemExit(1)
    digests[v] = json.loads(line)
same = {k: digests["before"][k] == digests["pairs"][k] for k in digests["before"] if not k.startswith("us_per_step")}
print("bitwise equal:", all(same.values()), json.dumps(same))
times = {k: {v: digests[v][k] for v in digests} for k in digests["before"] if k.startswith("us_per_step")}
print("65 536 drones in aviaries of D, all force terms, 64-step rollout, us per env step:", json.dumps(times))
res = {}
WORK = [("stack8x8192_ext_240hz", ["--steps", "64", "--warmup", "64"]), ("stack8x8192_ext_240hz", ["--steps", "20", "--warmup", "5"]),
        ("stack8x8192_ext_pid_240hz", ["--steps", "64", "--warmup", "64"]), ("multihover2x16384_240hz", ["--steps", "64", "--warmup", "64"]),
        ("multihover2x16384_pid_240hz", ["--steps", "64", "--warmup", "64"]), ("hover65536_ext_240hz", ["--steps", "64", "--warmup", "64"]),
        ("hover65536_ext_pid_240hz", ["--steps", "64", "--warmup", "64"]), ("hover65536_240hz", ["--steps", "20", "--warmup", "5"])]
for rnd in range(2):
    for wl, extra in WORK:
        for mode in ("rollout", "graph"):
            if mode == "graph" and (extra[1] != "64" or "pid" in wl or wl == "hover65536_240hz"):      # (the single-step kernel's variants)
                continue
            for v, lib in LIBS.items():
                cmd = [sys.executable, os.path.join(R, "bench.py"), "--workload", wl, "--mode", mode, "--no-cpu-baseline", "--no-hbm-leg", "--no-parity", "--no-second-leg",
                       "--no-dropin-leg", "--min-time", "0.5"] + extra
                p = subprocess.run(cmd, env=dict(os.environ, GPD_LIB=os.path.join(R, lib)), capture_output=True, text=True, timeout=300)
                line = next((l for l in reversed(p.stdout.splitlines()) if l.startswith("{")), None)
                if not line:
                    print(wl, v, "FAILED", p.stderr[-300:], flush=True)
                    continue
                j = json.loads(line)
                key = f"{wl} {mode} K={extra[1]}"
                res.setdefault(key, {}).setdefault(v, []).append(j["ms_per_step"] * 1e3)
                print(f"round {rnd} {key:44s} {v:7s}: {j['ms_per_step'] * 1e3:.4f} us per step", flush=True)
print("\nus per env step (min .. max over rounds)")
for k, d in res.items():
    print(f"{k:46s} " + "   ".join(f"{v}: {min(x):.4f}..{max(x):.4f}" for v, x in d.items()))
os.makedirs(os.path.join(R, "gpurun_out", "r06u"), exist_ok=True)
json.dump({"bitwise_equal": same, "us_per_env_step_by_aviary_size": times, "us_per_env_step": res}, open(os.path.join(R, "gpurun_out", "r06u", "ab_unroll.json"), "w"), indent=1)
