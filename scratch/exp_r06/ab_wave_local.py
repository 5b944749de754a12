#!/usr/bin/env python3
"""Rollouts of aviaries whose size does not divide 64: gpd_rollout1_kernel with whole aviaries per wave (GPD_ROLLOUT_WAVE_LOCAL=1) against
the compute-wave + store-wave kernel (=0, the rule up to round 6).  (1) digests of rollouts + final state: equal bit for bit;
(2) time per env step of a 64-step rollout of ~65 536 drones, all force terms, by aviary size."""
import hashlib, json, os, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SIZES = tuple(int(x) for x in os.environ.get("GPD_AB_SIZES", "3,5,6,7,9,10,12,15,20,21,22,24,28,31,33,40,48,63").split(","))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, R)
    import numpy as np, torch, bench
    from gym_pybullet_drones_amd.envs import VectorAviary
    from gym_pybullet_drones_amd.utils.enums import ActionType
    dev = torch.device("cuda", 0)
    out = {}
    for D in SIZES:
        for act, S, E in (("rpm", 1, 1000 // D + 3), ("pid", 2, 517 // D + 1)):
            rng = np.random.default_rng(D)
            xyz, rpy = bench.stack_scene(rng, E, D)
            env = VectorAviary(E, D, initial_xyzs=xyz, initial_rpys=rpy, physics=7, pyb_freq=240, ctrl_freq=240 // S, act=ActionType(act),
                               task="multihover", auto_reset=True, track_rpm=True, device=dev)
            env.core._cfg.trunc_counter = 30
            g = torch.Generator(device=dev); g.manual_seed(5)
            a = torch.rand((40, E, D, env.ACT_DIM), generator=g, device=dev) * 2 - 1
            if act == "pid":
                a = a * 0.5; a[..., 2] += 1.0
            h = hashlib.sha256()
            for _ in range(2):
                o, r, te, tr = env.core.rollout(a.contiguous(), update_latest=False)
                for t in (o, r, te, tr, env.core.kin_store, env.core.step_counter, env.core.last_rpm):
                    h.update(t.cpu().numpy().tobytes())
            out[f"D{D}_{act}_S{S}"] = h.hexdigest()[:16]
        E = 65536 // D
        rng = np.random.default_rng(D)
        xyz, rpy = bench.stack_scene(rng, E, D)
        env = VectorAviary(E, D, initial_xyzs=xyz, initial_rpys=rpy, physics=7, pyb_freq=240, ctrl_freq=240, act=ActionType("rpm"),
                           task="multihover", auto_reset=True, track_rpm=True, device=dev)
        a = (torch.rand((64, E, D, 4), device=dev) * 2 - 1).contiguous()
        for _ in range(3):
            env.core.rollout(a, update_latest=False)
        out[f"us_per_step_D{D}"] = bench.event_seconds(lambda: env.core.rollout(a, update_latest=False), 20) * 1e6 / 64
    print(json.dumps(out))
    raise SystemExit(0)
res = {}
for how in ("0", "1"):
    p = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=dict(os.environ, GPD_ROLLOUT_WAVE_LOCAL=how), capture_output=True, text=True)
    line = next((l for l in p.stdout.splitlines() if l.startswith("{")), None)
    if line is None:
        print(how, "FAILED", p.stderr[-1500:]); raise SystemExit(1)
    res[how] = json.loads(line)
same = {k: res["0"][k] == res["1"][k] for k in res["0"] if not k.startswith("us_")}
print("bitwise equal:", all(same.values()), [k for k, v in same.items() if not v])
rows = {D: {"lanes_with_a_drone_per_wave": 64 // D * D, "store_wave_kernel_us": res["0"][f"us_per_step_D{D}"], "wave_local_us": res["1"][f"us_per_step_D{D}"]} for D in SIZES}
for D, r in rows.items():
    print(f"D {D:3d}  lanes {r['lanes_with_a_drone_per_wave']:2d}  store-wave kernel {r['store_wave_kernel_us']:7.3f}  wave-local {r['wave_local_us']:7.3f}  ratio {r['wave_local_us'] / r['store_wave_kernel_us']:.2f}")
os.makedirs(os.path.join(R, "gpurun_out", "r06s"), exist_ok=True)
json.dump({"bitwise_equal": same, "us_per_env_step": rows}, open(os.path.join(R, "gpurun_out", "r06s", "ab_wave_local.json"), "w"), indent=1)
