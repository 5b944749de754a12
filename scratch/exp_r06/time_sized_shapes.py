#!/usr/bin/env python3
"""What each class of compile-time variant of gpd_rollout1_kernel buys, shape by shape: 64-step rollouts of 65 536 drones timed with
HIP events in two processes, GPD_ROLLOUT_SIZED=0 (generic kernels) and the default.  -> gpurun_out/r06y/time_sized_shapes.json"""
import json
import os
import subprocess
import sys

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
# (drones per aviary, physics flags, ctrl_freq, act): the variant each one takes is in the comment of step_rollout.hip's dispatch
SHAPES = [(2, 0, 30, "rpm"), (2, 0, 240, "rpm"), (2, 4, 30, "rpm"), (2, 7, 240, "rpm"), (3, 0, 240, "rpm"), (3, 4, 240, "rpm"), (4, 7, 240, "rpm"),
          (5, 0, 240, "pid"), (8, 0, 240, "rpm"), (8, 4, 240, "rpm"), (16, 7, 240, "rpm"), (1, 7, 240, "one_d_rpm"), (1, 1, 240, "rpm"),
          # with the ground plane (8) / Bullet's damping (16) bits a `Physics.PYB_*` member adds by default: the HI variants
          (8, 15, 240, "rpm"), (2, 12, 240, "rpm"), (1, 15, 240, "rpm"), (4, 15, 240, "rpm"), (8, 31, 240, "pid"),
          # single drones with the ground plane alone (`Physics.PYB`'s default mask), at 240 and at 30 Hz control
          (1, 8, 240, "rpm"), (1, 8, 30, "one_d_rpm"), (1, 24, 30, "pid"), (2, 8, 30, "rpm"), (2, 8, 240, "rpm"),
          # the add-on sets at the reference's default 30 Hz control
          (1, 15, 30, "rpm"), (1, 7, 30, "pid"), (2, 12, 30, "rpm"), (8, 15, 30, "rpm"), (4, 7, 30, "rpm")]
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, R)
    import numpy as np
    import torch
    import bench
    from gym_pybullet_drones_amd.envs import VectorAviary
    from gym_pybullet_drones_amd.utils.enums import ActionType
    dev = torch.device("cuda", 0)
    out = {}
    for D, phys, ctrl, act in SHAPES:
        E = 65536 // D
        xyz, rpy = bench.stack_scene(np.random.default_rng(D), E, D)
        env = VectorAviary(E, D, initial_xyzs=xyz, initial_rpys=rpy, physics=phys, pyb_freq=240, ctrl_freq=ctrl, act=ActionType(act),
                           task="hover" if D == 1 else "multihover", auto_reset=True, track_rpm=True, device=dev)
        a = (torch.rand((64, E, D, env.ACT_DIM), device=dev) * 2 - 1)
        if act == "pid":
            a = a * 0.3
            a[..., 2] += 1.0
        a = a.contiguous()
        for _ in range(3):
            env.core.rollout(a, update_latest=False)
        best = min(bench.event_seconds(lambda: env.core.rollout(a, update_latest=False), 40) for _ in range(3))
        out[f"D{D}_flags{phys}_{ctrl}hz_{act}"] = best * 1e6 / 64
    print(json.dumps(out))
    raise SystemExit(0)
res = {}
for rnd in range(2):
    for v, e in (("generic", {"GPD_ROLLOUT_SIZED": "0"}), ("sized", {})):
        p = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=dict(os.environ, **e), capture_output=True, text=True)
        line = next((l for l in p.stdout.splitlines() if l.startswith("{")), None)
        if line is None:
            print(v, "FAILED", p.stderr[-800:])
            raise SystemExit(1)
        for k, t in json.loads(line).items():
            res.setdefault(k, {}).setdefault(v, []).append(t)
print("us per env step, 64-step rollouts of 65 536 drones (best of 3 x 40 launches, two rounds)")
for k, d in res.items():
    g, s = min(d["generic"]), min(d["sized"])
    print(f"{k:28s} generic {g:7.4f}   sized {s:7.4f}   {100 * (s / g - 1):+5.1f} %")
os.makedirs(os.path.join(R, "gpurun_out", "r06y"), exist_ok=True)
json.dump(res, open(os.path.join(R, "gpurun_out", "r06y", "time_sized_shapes.json"), "w"), indent=1)
