"""Which CU did every workgroup of a gpd_rollout launch run on, and how long did its step loop take?  (variant build with
-DGPD_EXP_TS -DGPD_EXP_HWID).  usage: GPD_LIB=scratch/exp/libgpd_hw.so python scratch/launch_hwid.py [K]"""
import ctypes, sys, collections
import numpy as np
import torch
sys.path.insert(0, ".")
import bench
K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda:0")
env = bench.make_env(bench.WORKLOADS["hover65536_240hz"], dev, 0)
core = env.core
acts = torch.rand((K, core.N, 4), device=dev) * 2 - 1
for _ in range(12):
    core.rollout(acts, update_latest=False)
torch.cuda.synchronize()
ts = np.zeros((8, 4096, 4), dtype=np.uint64)
cnt = np.zeros(4096, dtype=np.uint32)
core.lib.gpd_debug_ts(ts.ctypes.data_as(ctypes.c_void_p), cnt.ctypes.data_as(ctypes.c_void_p))
W = 256
for slot in range(3):
    hw = ts[slot, :W, 1]
    xcc = (hw >> np.uint64(32)) & np.uint64(0xf)
    h = hw & np.uint64(0xffffffff)
    cu, sh, se = (h >> np.uint64(8)) & np.uint64(0xf), (h >> np.uint64(12)) & np.uint64(1), (h >> np.uint64(13)) & np.uint64(7)
    dur = (ts[slot, :W, 3].astype(np.float64) - ts[slot, :W, 0].astype(np.float64)) * 0.01
    where = [(int(x), int(e), int(s_), int(c)) for x, e, s_, c in zip(xcc, se, sh, cu)]
    per_cu = collections.Counter(where)
    share = np.array([per_cu[w] for w in where])
    print(f"launch slot {slot}: {len(per_cu)} distinct CUs for {W} workgroups; workgroups per CU histogram {sorted(collections.Counter(per_cu.values()).items())}")
    for k in sorted(set(share)):
        print(f"   workgroups on a CU holding {k}: n={int((share == k).sum())}, duration mean {dur[share == k].mean():.2f} us, min {dur[share == k].min():.2f}, max {dur[share == k].max():.2f}")
    print("   per XCC mean duration:", {int(x): round(float(dur[xcc == x].mean()), 2) for x in sorted(set(xcc))}, "WGs per XCC:", sorted(collections.Counter(int(x) for x in xcc).items()))
d = (ts[:, :W, 3].astype(np.float64) - ts[:, :W, 0].astype(np.float64)) * 0.01
loop = (ts[:, :W, 2].astype(np.float64) - ts[:, :W, 0].astype(np.float64)) * 0.01
print("correlation of a workgroup's duration between launches:", np.round(np.corrcoef(d[:6])[0, 1:6], 2))
same_cu = [(ts[0, :W, 1] == ts[k, :W, 1]).mean() for k in range(1, 6)]
print("fraction of workgroups on the same CU as in launch slot 0:", np.round(same_cu, 2))
print("mean over launches per workgroup: min %.2f max %.2f std %.2f; std within a launch %.2f" % (d.mean(0).min(), d.mean(0).max(), d.mean(0).std(), d.std(1).mean()))
print("max - mean per launch:", np.round(d.max(1) - d.mean(1), 2))
