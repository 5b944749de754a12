#!/usr/bin/env python3
"""The hover4m rollout leg runs at ~0.62 OR ~0.72 of 8 TB/s -- per process, on one box (hbm_box_probe.sh).  Is it the placement of the
buffers?  One process, the environment and its rollout buffers allocated again and again (cache emptied in between), each time the
64-step launch timed and the device addresses printed."""
import gc
import os
import sys

import torch

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import bench  # noqa: E402

dev = torch.device("cuda:0")
w = bench.WORKLOADS["hover4m_240hz"]
keep = []
for trial in range(int(sys.argv[1]) if len(sys.argv) > 1 else 8):
    env = bench.make_env(w, dev, seed=1000)
    acts = bench.make_actions(w, env, dev, seed=2000, pool=64)
    out = bench.launch_rollout(env, acts, 64)
    for _ in range(3):
        bench.launch_rollout(env, acts, 64)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(60):
        bench.launch_rollout(env, acts, 64)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 60
    bytes_launch = env.core.bytes_per_rollout(64)
    print(f"trial {trial}: {us:8.1f} us per launch, {bytes_launch / us / 1e3 / 8000:.3f} of 8 TB/s | obs {out[0].data_ptr():#x} actions {acts.data_ptr():#x} "
          f"reward {out[1].data_ptr():#x} state {env.core.kin_store.data_ptr():#x}", flush=True)
    if trial % 3 == 2:
        keep.append(torch.empty(int(1.5e9), dtype=torch.uint8, device=dev))       # shift what the next trial gets
    del env, acts, out
    gc.collect()
    torch.cuda.empty_cache()
