"""In-kernel timeline of back-to-back gpd_rollout launches (variant build with -DGPD_EXP_TS): where the time of a launch goes
outside its step loop.  usage: GPD_LIB=scratch/exp/libgpd_ts.so python scratch/launch_timeline.py [K]"""
import ctypes, sys
import numpy as np
import torch
sys.path.insert(0, ".")
import bench
K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda:0")
env = bench.make_env(bench.WORKLOADS["hover65536_240hz"], dev, 0)
core = env.core
acts = torch.rand((K, core.N, 4), device=dev) * 2 - 1
if len(sys.argv) > 2 and sys.argv[2] == "hover":      # no rare block is ever taken: what is left of the spread is the hardware's
    acts.zero_()
for _ in range(3):
    core.rollout(acts, update_latest=False)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream(dev)
with torch.cuda.stream(s):
    with torch.cuda.graph(g, stream=s):
        for _ in range(6):
            core.rollout(acts, update_latest=False)
for _ in range(5):
    g.replay()
torch.cuda.synchronize()
lib = core.lib
ts = np.zeros((8, 4096, 4), dtype=np.uint64)
cnt = np.zeros(4096, dtype=np.uint32)
lib.gpd_debug_ts(ts.ctypes.data_as(ctypes.c_void_p), cnt.ctypes.data_as(ctypes.c_void_p))
W = 256
t = ts[:, :W, :].astype(np.float64) * 0.01     # 100 MHz -> us
order = np.argsort(t[:, 0, 0])                 # launches in time order (slot ring)
t = t[order]
base = t.min()
for i in range(8):
    a = t[i] - base
    line = f"launch {i}: start {a[:,0].min():8.2f}..{a[:,0].max():8.2f} | state in regs {a[:,1].min():8.2f}..{a[:,1].max():8.2f} | loop done {a[:,2].min():8.2f}..{a[:,2].max():8.2f} | stores done {a[:,3].min():8.2f}..{a[:,3].max():8.2f}"
    if i:
        line += f" | gap after previous launch's last store {a[:,0].min() - (t[i-1][:,3].max() - base):6.2f}"
    print(line)
d = t[1:7]
print(f"K={K}: per launch (median over launches, us): first start -> last start {np.median(d[:,:,0].max(1)-d[:,:,0].min(1)):.2f}; prologue (start -> state in regs, median WG) {np.median(d[:,:,1]-d[:,:,0]):.2f}; "
      f"loop {np.median(d[:,:,2]-d[:,:,1]):.2f}; epilogue {np.median(d[:,:,3]-d[:,:,2]):.2f}; launch period {np.median(np.diff(t[:,:,0].min(1))):.2f}; "
      f"last store -> next first start {np.median(t[1:,:,0].min(1) - t[:-1,:,3].max(1)):.2f}")
