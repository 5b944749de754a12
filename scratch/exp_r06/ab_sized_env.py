#!/usr/bin/env python3
"""A/B of the rollout kernel's compile-time variants inside ONE library: GPD_ROLLOUT_SIZED=0 (the generic kernels) against the default.
(1) digests of rollouts + single steps, single drones at 30 Hz control (eight sub-steps: the NS = 8 variant) for every action type, with
and without the force terms; (2) bench lines, interleaved.   usage: ab_sized_env.py [workload:K ...]
(First use: a variant with the sub-step count as a compile-time 8 -- `NS`, the reference's default 30 Hz control -- bitwise equal and
2.5 % faster (1.856-1.865 -> 1.810-1.814 us per env step at 65 536 drones): fourteen more fully unrolled kernels for that; not kept,
profiles/r06_ab_substeps8_variant.json.)"""
import hashlib
import json
import os
import subprocess
import sys

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, R)
    import torch
    from gym_pybullet_drones_amd.envs import VectorAviary
    from gym_pybullet_drones_amd.utils.enums import ActionType
    out = {}
    dev = torch.device("cuda", 0)
    for act in ("rpm", "pid", "vel", "one_d_rpm", "one_d_pid"):
        for phys in (0, 7, 8):
            for ctrl in (30, 48):
                E = 1000
                env = VectorAviary(E, 1, physics=phys, pyb_freq=240, ctrl_freq=ctrl, act=ActionType(act), task="hover", auto_reset=True, track_rpm=True, device=dev)
                g = torch.Generator(device=dev); g.manual_seed(5)
                a = torch.rand((64, E, 1, env.ACT_DIM), generator=g, device=dev) * 2 - 1
                if act in ("pid", "one_d_pid"):
                    a = a * 0.5
                    a[..., -1] += 1.0
                if act == "vel":
                    a[..., 3] = a[..., 3].abs()
                h = hashlib.sha256()
                o, r, te, tr = env.core.rollout(a.contiguous(), update_latest=False)
                for t in (o, r, te, tr, env.core.kin_store):
                    h.update(t.cpu().numpy().tobytes())
                for k in range(4):
                    o, r, te, tr = env.core.step(a[k].contiguous())
                    for t in (o, r, te, tr):
                        h.update(t.cpu().numpy().tobytes())
                out[f"{act}_phys{phys}_{ctrl}hz"] = h.hexdigest()[:16]
    print(json.dumps(out))
    raise SystemExit(0)

ENVS = {"generic": {"GPD_ROLLOUT_SIZED": "0"}, "sized": {}}
dig = {}
for v, e in ENVS.items():
    p = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=dict(os.environ, **e), capture_output=True, text=True)
    line = next((l for l in p.stdout.splitlines() if l.startswith("{")), None)
    if line is None:
        print(v, "FAILED", p.stderr[-800:])
        raise SystemExit(1)
    dig[v] = json.loads(line)
same = {k: dig["generic"][k] == dig["sized"][k] for k in dig["generic"]}
print("bitwise equal:", all(same.values()), json.dumps(same))
WORK = [w.split(":") for w in sys.argv[1:]] or [["hover65536_30hz", "64"], ["hover65536_30hz", "20"], ["hover4096_30hz", "64"]]
res = {}
for rnd in range(3):
    for wl, K in WORK:
        for v, e in ENVS.items():
            cmd = [sys.executable, os.path.join(R, "bench.py"), "--workload", wl, "--no-cpu-baseline", "--no-hbm-leg", "--no-parity", "--no-second-leg", "--no-dropin-leg",
                   "--min-time", "0.5", "--steps", K, "--warmup", K]
            p = subprocess.run(cmd, env=dict(os.environ, **e), capture_output=True, text=True, timeout=300)
            line = next((l for l in reversed(p.stdout.splitlines()) if l.startswith("{")), None)
            if not line:
                print(wl, v, "FAILED", p.stderr[-300:], flush=True)
                continue
            j = json.loads(line)
            res.setdefault(f"{wl} K={K}", {}).setdefault(v, []).append(j["ms_per_step"] * 1e3)
            print(f"round {rnd} {wl} K={K:3s} {v:8s}: {j['ms_per_step'] * 1e3:.4f} us per step", flush=True)
print("\nus per env step (min .. max over rounds)")
for k, d in res.items():
    print(f"{k:40s} " + "   ".join(f"{v}: {min(x):.4f}..{max(x):.4f}" for v, x in d.items()))
os.makedirs(os.path.join(R, "gpurun_out", "r06v"), exist_ok=True)
json.dump({"bitwise_equal": same, "us_per_env_step": res}, open(os.path.join(R, "gpurun_out", "r06v", "ab_sized_env.json"), "w"), indent=1)
