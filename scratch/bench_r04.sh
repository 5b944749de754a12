#!/bin/bash
# The unprofiled bench lines of round 4 (one GPU-box call): written to gpurun_out/r04_bench_*.json, copied to profiles/ afterwards.
mkdir -p gpurun_out
run() { name=$1; shift; timeout 600 python bench.py "$@" 2> gpurun_out/r04_bench_$name.err | tail -1 > gpurun_out/r04_bench_$name.json
  python - gpurun_out/r04_bench_$name.json $name <<'PY'
import json, sys
try:
    j = json.load(open(sys.argv[1]))
except Exception as e:
    print(sys.argv[2], "NO LINE", e); sys.exit(0)
o = j.get("one_launch_per_step") or {}; p = j.get("parity") or {}; h = j.get("hbm_saturating") or {}
print(sys.argv[2], "us/step %.4f" % (j["ms_per_step"] * 1e3), "value %.4g" % j["value"], "bound", j["roofline"]["bound"], "frac", j["roofline"].get("frac"),
      ("| step leg %.3f us frac %.3f" % (o["us_per_step"], o["roofline"]["frac"])) if o else "", "| parity", p.get("max", p.get("force_max_abs_err_over_max_force")), p.get("ok"), p.get("ok_by"),
      ("env ratio %.2f" % p["envelope"]["ratio"]) if "envelope" in p else "", ("| hbm leg frac %.3f" % h["frac"]) if h else "")
PY
}
run driver_cmd --gpus 1 --steps 20 --warmup 5
run default --no-cpu-baseline
for wl in hover65536_30hz hover65536_pid_240hz hover65536_ext_240hz hover65536_ext_pid_240hz stack8x8192_ext_240hz stack8x8192_ext_pid_240hz multihover2x16384_240hz hover65536_240hz_termobs hover65536_30hz_history hover65536_30hz_policy hover4096_240hz hover4m_240hz; do
  run $wl --workload $wl --no-cpu-baseline --no-hbm-leg
done
run swarm65536_ext_240hz --workload swarm65536_ext_240hz --steps 240 --warmup 24
run swarm1m_ext_240hz --workload swarm1m_ext_240hz --steps 64 --warmup 16 --no-cpu-baseline
GPD_DIST_BACKEND=gloo GPD_BENCH_SINGLE_DEVICE=1 run two_ranks_one_device --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline --min-time 0.05
