"""Where a launch of gpd_swarm_step_kernel spends its time (a build with -DGPD_EXP_TS -DGPD_EXP_TSS: per-workgroup wall_clock64
stamps of thread 0 at entry / state in registers / physics + tail (reductions, barrier) done / every store issued / acknowledged).
usage: python scratch/build_variant.py scratch/exp_r04/libgpd_tss.so --define GPD_EXP_TS GPD_EXP_TSS
       GPD_LIB=$PWD/scratch/exp_r04/libgpd_tss.so python scratch/exp_r04/step_timeline.py [workload]"""
import ctypes, os, sys
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import torch
import bench
wl = sys.argv[1] if len(sys.argv) > 1 else "swarm65536_ext_240hz"
dev = torch.device("cuda:0")
env = bench.make_env(bench.WORKLOADS[wl], dev, seed=1000)
act = torch.full((env.NUM_DRONES, 4), float(env.HOVER_RPM), device=dev)
env.reset()
g = torch.cuda.CUDAGraph()
for i in range(20):
    env.step(act)
torch.cuda.synchronize()
with torch.cuda.graph(g):                    # back-to-back launches, as the bench times them
    for i in range(16):
        env.step(act)
for i in range(4):
    g.replay()
torch.cuda.synchronize()
ts = np.zeros((8, 4096, 4), dtype=np.uint64)
cnt = np.zeros(4096, dtype=np.uint32)
env.core.lib.gpd_debug_ts(ts.ctypes.data_as(ctypes.c_void_p), cnt.ctypes.data_as(ctypes.c_void_p))
W = min(4096, -(-env.NUM_DRONES // 256))
print(wl, "workgroups stamped:", W, "launches recorded per workgroup:", cnt[:3])
for slot in range(8):
    raw = ts[slot, :W]
    t = raw[:, :3].astype(np.float64) * 0.01                    # us (100 MHz)
    t3 = (raw[:, 3] & np.uint64(0xffffffff)).astype(np.float64)
    t4 = (raw[:, 3] >> np.uint64(32)).astype(np.float64)
    lo = (raw[:, 0] & np.uint64(0xffffffff)).astype(np.float64)
    # (the two packed stamps are the low words: differences against the entry stamp's low word)
    d_issue, d_ack = ((t3 - lo) % 2 ** 32) * 0.01, ((t4 - lo) % 2 ** 32) * 0.01
    t0 = t[:, 0].min()
    print(f"slot {slot}: entry spread {t[:,0].max() - t0:.2f} | loaded +{np.median(t[:,1] - t[:,0]):.2f} (max {np.max(t[:,1] - t[:,0]):.2f}) | "
          f"physics + tail +{np.median(t[:,2] - t[:,1]):.2f} (max {np.max(t[:,2] - t[:,1]):.2f}) | stores issued +{np.median(d_issue - (t[:,2] - t[:,0])):.2f} | "
          f"acknowledged +{np.median(d_ack - d_issue):.2f} (max {np.max(d_ack - d_issue):.2f}) | workgroup lifetime median {np.median(d_ack):.2f}, "
          f"max {d_ack.max():.2f}; last ack after first entry {np.max(t[:,0] - t0 + d_ack):.2f}")
