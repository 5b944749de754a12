"""GPU-box debugging of the swarm bench path: eager steps with periodic synchronisation and sanity prints, then the
graph-replay pattern of bench.py."""
import faulthandler, os, sys
faulthandler.enable()
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device("cuda:0")
w = bench.WORKLOADS["swarm65536_ext_240hz"]
env = bench.make_env(w, dev, 1000)
acts = bench.make_actions(w, env, dev, 2000, 64)
N = env.NUM_DRONES
def report(tag):
    torch.cuda.synchronize()
    k = env.core.kin[:, :N]
    fin = torch.isfinite(k).all(dim=0)
    print(tag, "finite", int(fin.sum()), "z", float(k[2][fin].min()), float(k[2][fin].max()), "xy", float(k[0:2][:, fin].abs().max()),
          "dw", float(env.dw_force[:N].abs().max()), flush=True)
mode = sys.argv[1] if len(sys.argv) > 1 else "eager"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
if mode == "eager":
    for i in range(steps):
        env.step(acts[i % 64])
        if i % 256 == 255:
            report(f"step {i + 1}")
else:
    for i in range(8):
        env.step(acts[i % 64])
    report("warm")
    s = torch.cuda.Stream(dev); s.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(s):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for i in range(64):
                env.step(acts[i % 64])
    torch.cuda.current_stream(dev).wait_stream(s)
    for r in range(steps // 64):
        if r % 16 == 0:
            env.reset()
        g.replay()
        if r % 4 == 3:
            report(f"replay {r + 1}")
print("done")
