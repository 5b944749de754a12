"""Timing experiment (GPU box): what do the reward / flag streams of the headline rollout cost, and is it the DRAM side?
(a) the normal call; (b) env_step_stride = 0: every step's reward/flags overwrite ONE row (same instructions, no new DRAM pages)."""
import ctypes, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from gym_pybullet_drones_amd import _native
dev = torch.device("cuda:0")
w = bench.WORKLOADS["hover65536_240hz"]
env = bench.make_env(w, dev, 1000)
acts = bench.make_actions(w, env, dev, 2000, 64)
c = env.core
K = 64
obs, rew, term, trunc, _ = c._rollout_buffers(K)
P = lambda t: ctypes.c_void_p(t.data_ptr())
def call(e_stride):
    rc = c.lib.gpd_rollout(ctypes.byref(c._params), ctypes.byref(c._state), ctypes.byref(c._cfg), K, P(acts), c.N * c.A, P(c.target), P(c.init_pose),
                           P(obs), c.N * 12, P(rew), P(term), P(trunc), e_stride, None, ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
    assert rc == 0
for name, es in (("normal (stride E)", c.E), ("stride 0 (one row)", 0), ("normal (stride E)", c.E), ("stride 0 (one row)", 0)):
    for _ in range(20): call(es)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(2000): call(es)
    e1.record(); torch.cuda.synchronize()
    print(f"{name:22s} {e0.elapsed_time(e1) * 1e3 / (2000 * K):.4f} us per step")
