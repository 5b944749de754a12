"""Time gpd_downwash_global alone on the swarm bench scene (HIP events, 300 calls).  GPD_LIB selects the build."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
import bench
dev = torch.device("cuda:0")
env = bench.make_env(bench.WORKLOADS["swarm65536_ext_240hz"], dev, 0)
env.reset()
for _ in range(20):
    env.downwash()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(300):
    env.downwash()
b.record()
torch.cuda.synchronize()
print(f"{a.elapsed_time(b) / 300 * 1e3:.2f} us per gpd_downwash_global call (count + scan + scatter + force)")
