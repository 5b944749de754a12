"""Where a replay launch of dwg_force_kernel<2> spends its time (a build with -DGPD_EXP_TS -DGPD_EXP_TSF: per-workgroup
wall_clock64 stamps at entry / set-up done / first tile staged / evaluation done).
usage: GPD_LIB=scratch/exp/libgpd_tsf.so python scratch/exp_r03/force_timeline.py"""
import ctypes, os, sys
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import torch
import bench
dev = torch.device("cuda:0")
env = bench.make_env(bench.WORKLOADS["swarm65536_ext_240hz"], dev, seed=1000)
act = torch.full((env.NUM_DRONES, 4), float(env.HOVER_RPM), device=dev)
env.reset()
for i in range(45):
    env.step(act)
torch.cuda.synchronize()
ts = np.zeros((8, 4096, 4), dtype=np.uint64)
cnt = np.zeros(4096, dtype=np.uint32)
env.core.lib.gpd_debug_ts(ts.ctypes.data_as(ctypes.c_void_p), cnt.ctypes.data_as(ctypes.c_void_p))
W = 1024
print("replay launches recorded per workgroup:", cnt[:4], "...")
for slot in range(8):
    t = ts[slot, :W].astype(np.float64) * 0.01            # us (100 MHz)
    if not t[:, 0].any():
        continue
    t0 = t[:, 0].min()
    rel = t - t0
    d = np.diff(t, axis=1)
    print(f"slot {slot}: entry {rel[:,0].min():.2f}..{rel[:,0].max():.2f} | set-up done {np.median(rel[:,1]):.2f} (max {rel[:,1].max():.2f}) | "
          f"tile staged {np.median(rel[:,2]):.2f} (max {rel[:,2].max():.2f}) | evaluated {np.median(rel[:,3]):.2f} (max {rel[:,3].max():.2f}) || "
          f"phases median: set-up {np.median(d[:,0]):.2f}, staging {np.median(d[:,1]):.2f}, evaluation {np.median(d[:,2]):.2f}; max {d[:,0].max():.2f} {d[:,1].max():.2f} {d[:,2].max():.2f}")
nb = env._pair_nb.cpu().numpy().astype(np.int64) & 0xffff          # [groups][4][16]
per_wave = nb.sum(axis=2)[:W]                                       # batches per wave
t = ts[7, :W].astype(np.float64) * 0.01
ev = t[:, 3] - t[:, 2]
print("batches per wave: mean %.1f, median %.1f, max %d; per group (max over its waves): median %.1f, max %d" %
      (per_wave.mean(), np.median(per_wave), per_wave.max(), np.median(per_wave.max(axis=1)), per_wave.max()))
print("tiles per group with batches:", np.bincount((nb[:W] > 0).any(axis=1).sum(axis=1)))
for lo, hi in ((0, 4), (4, 5), (5, 6), (6, 7), (7, 8), (8, 10), (10, 99)):
    m = (ev >= lo) & (ev < hi)
    if m.any():
        print(f"  evaluation {lo}-{hi} us: {m.sum():4d} groups, batches of wave 0 (as timed) mean {per_wave[m, 0].mean():.1f}, max over waves mean {per_wave[m].max(axis=1).mean():.1f}")
print("correlation(evaluation time, batches of wave 0):", np.corrcoef(ev, per_wave[:, 0])[0, 1])
hw = None
print("margin chosen by the last binning:", float(env._drift[2]), "of", env.list_delta, "; dmax now:", float(env.pos4[env.NUM_DRONES:, 3].max()) ** 0.5)
for k in range(200):
    env.step(act)
    if env._since_bin == env.rebin_every - 1:
        print(f"  step {46 + k}: margin {float(env._drift[2]):.4f}, dmax just before the binning {float(env.pos4[env.NUM_DRONES:, 3].max()) ** 0.5:.4f}, "
              f"batches per wave {((env._pair_nb.cpu().numpy().astype(np.int64) & 0xffff).sum(axis=2)[:W]).mean():.1f}")
