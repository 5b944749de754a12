"""Where a replay launch of dwg_force_kernel<2> spends its time with the round-4 gather replay (a build with -DGPD_EXP_TS
-DGPD_EXP_TSF: thread 0 of every workgroup stamps entry / set-up done (every set-up load waited for) / evaluation done).
usage: python scratch/build_variant.py scratch/exp_r04/libgpd_tsf.so --define GPD_EXP_TS GPD_EXP_TSF
       GPD_LIB=$PWD/scratch/exp_r04/libgpd_tsf.so python scratch/exp_r04/force_timeline.py"""
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
for i in range(20):
    env.step(act)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    for i in range(16):
        env.step(act)
for i in range(3):
    g.replay()
torch.cuda.synchronize()
ts = np.zeros((8, 4096, 4), dtype=np.uint64)
cnt = np.zeros(4096, dtype=np.uint32)
env.core.lib.gpd_debug_ts(ts.ctypes.data_as(ctypes.c_void_p), cnt.ctypes.data_as(ctypes.c_void_p))
W = 1024
nb = (env._pair_nb.cpu().numpy().astype(np.int64) & 0xffff)[:W, :, 0]              # [group][wave]: the wave's batches
print("replay launches recorded per workgroup:", cnt[:3], "| batches per wave: mean %.1f, max %d" % (nb.mean(), nb.max()))
for slot in range(8):
    t = ts[slot, :W].astype(np.float64) * 0.01
    if not t[:, 0].any():
        continue
    t0 = t[:, 0].min()
    setup, ev = t[:, 1] - t[:, 0], t[:, 3] - t[:, 1]
    print(f"slot {slot}: entry spread {t[:,0].max() - t0:.2f} | set-up {np.median(setup):.2f} (max {setup.max():.2f}) | evaluation {np.median(ev):.2f} "
          f"(p95 {np.percentile(ev, 95):.2f}, max {ev.max():.2f}) | workgroup lifetime median {np.median(t[:,3] - t[:,0]):.2f}, max {np.max(t[:,3] - t[:,0]):.2f} | "
          f"last workgroup done {np.max(t[:,3]) - t0:.2f} after the first entry | corr(evaluation, batches of wave 0) {np.corrcoef(ev, nb[:, 0])[0, 1]:.2f}")
