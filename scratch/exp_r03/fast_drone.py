"""What ONE fast drone costs a hovering swarm of 65 536: the bench scene, K drones at 3 m/s sideways, a second of flight stepped
eagerly (GPU-bound: 13 us of host time per step).  usage: [GPD_LIB=...] python scratch/exp_r03/fast_drone.py [K ...]"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import torch
import bench
dev = torch.device("cuda:0")
env = bench.make_env(bench.WORKLOADS["swarm65536_ext_240hz"], dev, seed=1000)
act = torch.full((env.NUM_DRONES, 4), float(env.HOVER_RPM), device=dev)
for K in [int(a) for a in sys.argv[1:]] or [0, 1, 16]:
    res = []
    for rep in range(3):
        env.reset()
        if K:
            kin = env.core.kin[:, :env.NUM_DRONES].clone()
            kin[7, torch.arange(K, device=dev) * 4001 % env.NUM_DRONES] = 3.0
            env.core.set_state(kin=kin)
            env.invalidate() if hasattr(env, "invalidate") else None
        for _ in range(16):
            env.step(act)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(224):
            env.step(act)
        e1.record()
        torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) * 1e3 / 224)
    print(f"{os.environ.get('GPD_LIB', 'in-tree')[-24:]:>24}: {K:3d} drones at 3 m/s: {min(res):.2f} us per step (best of 3; margin at the end {float(env._drift[2]):.3f} m)")
