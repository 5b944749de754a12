#!/usr/bin/env python3
"""Round 6, after gpd_rollout1_kernel got its variants compiled for one aviary size / one flag set: the instruction and traffic counters
of the three BASELINE shapes those variants serve -- config 3 (i) `hover65536_ext_240hz`, 3 (ii) `stack8x8192_ext_240hz`, 5 per GPU
`multihover2x16384_240hz` -- with scratch/profile_r05.py's machinery: separate `rocprofv3 --kernel-trace --pmc` passes (FETCH_SIZE,
WRITE_SIZE, SQ1, SQ2) and one `--kernel-trace --stats` pass of each bench line.  ON THE GPU BOX -> gpurun_out/r06s/sized_counters.json:
per key the entries of profiles/kernel_counters.json and profiles/hbm_traffic.json, ready to merge (`python pmc_sized.py merge`)."""
import glob
import importlib.util
import json
import os
import shutil
import sys

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
OUTF = os.path.join(R, "gpurun_out", "r06s", "sized_counters.json")
WORK = ("stack8x8192_ext_240hz", "multihover2x16384_240hz", "hover65536_ext_240hz")

if len(sys.argv) > 1 and sys.argv[1] == "merge":
    res = json.load(open(OUTF))
    for name, part in (("kernel_counters.json", "counters"), ("hbm_traffic.json", "traffic")):
        f = os.path.join(R, "profiles", name)
        d = json.load(open(f))
        for key, rec in res.items():
            d[key] = rec[part]
        json.dump(d, open(f, "w"), indent=1)
    for key, rec in res.items():
        print(key, rec["counters"]["slots_per_wave_env_step"], rec["traffic"]["traffic_bytes"] / rec["traffic"]["algorithmic_bytes"], rec["traffic"]["rocprof_kernel_avg_ns"])
    raise SystemExit(0)

sys.argv = [sys.argv[0], "none"]                         # (profile_r05: measure nothing on import, keep its helpers)
spec = importlib.util.spec_from_file_location("p5", os.path.join(R, "scratch", "profile_r05.py"))
p5 = importlib.util.module_from_spec(spec)
spec.loader.exec_module(p5)
kern, spl = "gpd_rollout", 64
res = {}
for wl in WORK:
    rec = {}
    for gname, ctrs in (("FETCH_SIZE", ["FETCH_SIZE"]), ("WRITE_SIZE", ["WRITE_SIZE"]), ("SQ1", p5.SQ1), ("SQ2", p5.SQ2)):
        tag = f"r06_pmc_{wl}_rollout64_{gname}"
        args = ["--workload", wl, "--mode", "rollout", "--no-cpu-baseline", "--no-second-leg", "--no-hbm-leg", "--no-parity", "--steps", "64", "--warmup", "64", "--min-time", "0.002"]
        p5.rocprof(tag, ["--kernel-trace", "--pmc"] + ctrs, args)
        for c, v in p5.counters(tag, kern).items():
            rec.setdefault(c, v)
        if gname == "FETCH_SIZE":
            dd = [e - s for n, s, e, g in p5.dispatches(tag) if kern in n]
            dd = dd[len(dd) // 4:]
            if dd:
                rec["kernel_avg_ns_in_pmc_pass"] = sum(dd) / len(dd)
    tag = f"r06_trace_{wl}"
    line = p5.rocprof(tag, ["--kernel-trace", "--stats"], ["--workload", wl, "--no-cpu-baseline", "--no-second-leg", "--steps", "64", "--warmup", "64"])
    rows = [r for r in p5.by_grid(tag) if "rollout1" in r["Name"]]
    g = lambda c: rec.get(c, {}).get("mean_per_dispatch") or 0.0
    waves = g("SQ_WAVES")
    per = lambda c: g(c) / waves / spl
    slots = ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_SMEM")
    counters = {"slots_per_wave_env_step": sum(per(c) for c in slots), "valu_per_wave_env_step": per("SQ_INSTS_VALU"), "salu_per_wave_env_step": per("SQ_INSTS_SALU"),
                "branch_per_wave_env_step": per("SQ_INSTS_BRANCH"), "lds_per_wave_env_step": per("SQ_INSTS_LDS"),
                "vmem_per_wave_env_step": per("SQ_INSTS_VMEM_RD") + per("SQ_INSTS_VMEM_WR"), "wave_quad_cycles_per_env_step": per("SQ_WAVE_CYCLES"),
                "active_quad_cycles_per_env_step": per("SQ_ACTIVE_INST_ANY"), "parked_quad_cycles_per_env_step": per("SQ_WAIT_ANY"),
                "issue_stall_quad_cycles_per_env_step": per("SQ_WAIT_INST_ANY"), "waves": waves, "measured_in_round": 6,
                "kernel": rows[0]["Name"].split("(float*")[0] if rows else None}
    alg = line["roofline"]["bytes_per_launch"] if line else None
    traffic = {"env_steps_per_launch": spl, "algorithmic_bytes": alg, "measured_in_round": 6, "FETCH_SIZE_KB": g("FETCH_SIZE"), "WRITE_SIZE_KB": g("WRITE_SIZE"),
               # gfx950: FETCH_SIZE counts 32-byte... the guide's correction as scratch/refresh_profiles_r05.py applies it: 2 x FETCH_SIZE KB + WRITE_SIZE KB
               "traffic_bytes": 2 * g("FETCH_SIZE") * 1024 + g("WRITE_SIZE") * 1024,
               "rocprof_kernel_avg_ns": rows[0]["AverageNs"] if rows else None, "kernel_avg_ns_in_pmc_pass": rec.get("kernel_avg_ns_in_pmc_pass")}
    res[f"{wl}:rollout64"] = {"counters": counters, "traffic": traffic, "trace_by_grid": rows[:3],
                              "bench_line_under_rocprofv3": {k: line[k] for k in ("value", "ms_per_step", "roofline", "parity") if line and k in line}}
    print(wl, "slots", counters["slots_per_wave_env_step"], "valu", counters["valu_per_wave_env_step"], "traffic/alg", traffic["traffic_bytes"] / alg if alg else None,
          "trace ns", traffic["rocprof_kernel_avg_ns"], flush=True)
    os.makedirs(os.path.dirname(OUTF), exist_ok=True)
    json.dump(res, open(OUTF, "w"), indent=1)
    for f in glob.glob(os.path.join(p5.OUT, f"r06_pmc_{wl}_rollout64_*.csv")):
        shutil.copy(f, os.path.dirname(OUTF))
    for f in glob.glob(os.path.join(p5.OUT, f"r06_trace_{wl}_kernel_stats_by_grid.csv")):
        shutil.copy(f, os.path.dirname(OUTF))
for d in glob.glob(os.path.join(p5.OUT, "*")):
    if os.path.isdir(d):
        shutil.rmtree(d, ignore_errors=True)
