#!/usr/bin/env python3
"""Where the Python side of `HoverAviary().step()` spends its time (the library call replaced by a no-op)."""
import os, sys, time, warnings
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import torch
from gym_pybullet_drones_amd.envs import HoverAviary
from gym_pybullet_drones_amd.utils.enums import ActionType
warnings.simplefilter("ignore")
env = HoverAviary(act=ActionType.ONE_D_RPM)
env.reset(seed=0)
acts = np.random.default_rng(0).uniform(-1, 1, size=(4000, 1, 1)).astype(np.float32)
def per(fn, n=4000):
    t = time.perf_counter()
    for k in range(n):
        fn(k)
    return (time.perf_counter() - t) / n * 1e6
full = per(lambda k: env.step(acts[k]))
core = env._core
real = core.step_host
core.step_host = lambda: None
nolib = per(lambda k: env.step(acts[k]))
core.step_host = real
print(f"env.step {full:.2f} us; with the library call a no-op {nolib:.2f} us; step_host alone {per(lambda k: real()):.2f} us")
a = np.asarray(acts[0])
print(f"  _recordAction {per(lambda k: env._recordAction(a)):.2f}")
print(f"  action row write {per(lambda k: env._action_row.__setitem__(Ellipsis, np.reshape(a, env._action_row.shape))):.2f}")
print(f"  _updateAndStoreKinematicInformation {per(lambda k: env._updateAndStoreKinematicInformation()):.2f}")
print(f"  _computeObs {per(lambda k: env._computeObs()):.2f}  _computeReward {per(lambda k: env._computeReward()):.2f}  _computeTerminated {per(lambda k: env._computeTerminated()):.2f}  _computeTruncated {per(lambda k: env._computeTruncated()):.2f}  _computeInfo {per(lambda k: env._computeInfo()):.2f}")
v = env._host_views
print(f"  obs.astype {per(lambda k: v['obs'].astype(np.float64)):.2f}  quat.astype {per(lambda k: v['Q'].astype(np.float64)):.2f}  rates stack+astype {per(lambda k: np.stack([v['P'][:, 3], v['V'][:, 3], v['W']], axis=1).astype(np.float64)):.2f}  rpm.T.astype {per(lambda k: v['rpm'].T.astype(np.float64)):.2f}  scalars {per(lambda k: (float(v['reward'][0]), bool(v['terminated'][0]), bool(v['truncated'][0]))):.2f}")
import ctypes
L = core.lib
print(f"  ctypes call with 12 pointer args (gpd_step_sync(NULL...) -> EINVAL) {per(lambda k: L.gpd_step_sync(None, None, None, None, None, None, None, None, None, None, None, None)):.2f}; gpd_abi_version() {per(lambda k: L.gpd_abi_version()):.2f}")
row = env._action_row
def w(k): row[...] = np.reshape(acts[k], row.shape)
x = np.zeros((1, 12)); 
def dummy(k):
    for _ in range(6): y = x.astype(np.float32)
print("combos:")
print(f"  step_host {per(lambda k: real()):.2f}")
print(f"  write row + step_host {per(lambda k: (w(k), real())):.2f}")
print(f"  step_host + refresh {per(lambda k: (real(), env._updateAndStoreKinematicInformation())):.2f}")
print(f"  write + step_host + refresh {per(lambda k: (w(k), real(), env._updateAndStoreKinematicInformation())):.2f}")
print(f"  step_host + 6 unrelated astype ({per(dummy):.2f} alone) {per(lambda k: (real(), dummy(k))):.2f}")
print(f"  write + step_host + refresh + obs + record {per(lambda k: (env._recordAction(acts[k]), w(k), real(), env._updateAndStoreKinematicInformation(), env._computeObs())):.2f}")
print(f"  env.step {per(lambda k: env.step(acts[k])):.2f}")
