import sys, time, torch
sys.path.insert(0, ".")
from gym_pybullet_drones_amd.envs import VectorHoverAviary
from gym_pybullet_drones_amd.utils.enums import ActionType
env = VectorHoverAviary(4096, act=ActionType.RPM, ctrl_freq=240, device="cuda:0")
a = torch.zeros((4096, 1, 4), device="cuda:0")
for _ in range(200): env.step(a)
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 5000
for _ in range(n): env.step(a)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("host time per env.step() %.2f us (incl. GPU drain %.2f us)" % ((t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6))
core = env.core
t0 = time.perf_counter()
for _ in range(n): core.step(a)
t1 = time.perf_counter()
torch.cuda.synchronize()
print("host time per core.step() %.2f us" % ((t1 - t0) / n * 1e6))
