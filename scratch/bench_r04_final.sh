#!/bin/bash
# The lines that changed after scratch/bench_r04.sh ran (hbm_saturating times 64-step launches; the one-world kernels; the suite).
mkdir -p gpurun_out
run() { name=$1; shift; timeout 900 python bench.py "$@" 2> gpurun_out/r04_bench_$name.err | tail -1 > gpurun_out/r04_bench_$name.json; python - gpurun_out/r04_bench_$name.json $name <<'PY'
import json, sys
try:
    j = json.load(open(sys.argv[1]))
except Exception as e:
    print(sys.argv[2], "NO LINE", e); sys.exit(0)
h = j.get("hbm_saturating") or {}; r = j["roofline"]
print(sys.argv[2], "us/step %.4f" % (j["ms_per_step"] * 1e3), "value %.4g" % j["value"], r["bound"], "frac", r.get("frac"), ("| hbm leg frac %.3f steps/launch %s" % (h["frac"], h.get("env_steps_per_launch"))) if h else "",
      "| suite", {k: (v.get("value"), v.get("error")) for k, v in (j.get("suite") or {}).items()} if j.get("suite") else "")
PY
}
run driver_cmd --gpus 1 --steps 20 --warmup 5
run default --no-cpu-baseline
run swarm65536_ext_240hz --workload swarm65536_ext_240hz --steps 240 --warmup 24
run swarm1m_ext_240hz --workload swarm1m_ext_240hz --steps 64 --warmup 16 --no-cpu-baseline
GPD_DIST_BACKEND=gloo GPD_BENCH_SINGLE_DEVICE=1 run two_ranks_one_device --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline --min-time 0.05
