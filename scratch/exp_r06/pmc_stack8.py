#!/usr/bin/env python3
"""Round 6: the instruction and traffic counters of `stack8x8192_ext_240hz:rollout64` again (the two-mates downwash changed that kernel's
instruction stream), with scratch/profile_r05.py's machinery: separate `rocprofv3 --kernel-trace --pmc` passes (SQ1, SQ2, FETCH_SIZE,
WRITE_SIZE) and one `--kernel-trace --stats` pass of the bench line.  ON THE GPU BOX -> gpurun_out/r06q/stack8_counters.json"""
import importlib.util
import json
import os
import sys

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = [sys.argv[0], "none"]                         # (profile_r05: measure nothing on import, keep its helpers)
spec = importlib.util.spec_from_file_location("p5", os.path.join(R, "scratch", "profile_r05.py"))
p5 = importlib.util.module_from_spec(spec)
spec.loader.exec_module(p5)
key, wl, kern, spl = "stack8x8192_ext_240hz:rollout64", "stack8x8192_ext_240hz", "gpd_rollout", 64
rec = {"env_steps_per_launch": spl, "kernel": kern}
for gname, ctrs in (("FETCH_SIZE", ["FETCH_SIZE"]), ("WRITE_SIZE", ["WRITE_SIZE"]), ("SQ1", p5.SQ1), ("SQ2", p5.SQ2)):
    tag = "r06_pmc_stack8_" + gname
    args = ["--workload", wl, "--mode", "rollout", "--no-cpu-baseline", "--no-second-leg", "--no-hbm-leg", "--no-parity", "--steps", "64", "--warmup", "64", "--min-time", "0.002"]
    p5.rocprof(tag, ["--kernel-trace", "--pmc"] + ctrs, args)
    for c, v in p5.counters(tag, kern).items():
        rec.setdefault(c, v)
    if gname == "FETCH_SIZE":
        dd = [e - s for n, s, e, g in p5.dispatches(tag) if kern in n]
        dd = dd[len(dd) // 4:]
        if dd:
            rec["kernel_avg_ns_in_pmc_pass"] = sum(dd) / len(dd)
line = p5.rocprof("r06_trace_stack8", ["--kernel-trace", "--stats"], ["--workload", wl, "--no-cpu-baseline"])
rows = p5.by_grid("r06_trace_stack8")
out = {"key": key, "pmc": rec, "trace_by_grid": rows[:6], "bench_line_under_rocprofv3": line}
os.makedirs(os.path.join(R, "gpurun_out", "r06q"), exist_ok=True)
json.dump(out, open(os.path.join(R, "gpurun_out", "r06q", "stack8_counters.json"), "w"), indent=1)
g = lambda c: rec.get(c, {}).get("mean_per_dispatch")
per = lambda c: (g(c) or 0.0) / g("SQ_WAVES") / spl
print("slots per wave-step", sum(per(c) for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_SMEM")), "valu", per("SQ_INSTS_VALU"))
print("traffic bytes", 2 * g("FETCH_SIZE") * 1024 + g("WRITE_SIZE") * 1024, "kernel avg ns", rec.get("kernel_avg_ns_in_pmc_pass"))
for r in rows[:3]:
    print(r["Name"][:60], r["Grid_Size_X"], r["Calls"], r["AverageNs"])
import glob, shutil
for d in glob.glob(os.path.join(p5.OUT, "*")):
    if os.path.isdir(d):
        shutil.rmtree(d, ignore_errors=True)
