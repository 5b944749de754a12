#!/bin/bash
# Bench lines of the non-headline workloads -> gpurun_out/bench_<workload>.json (refresh_profiles.py copies them to profiles/)
mkdir -p gpurun_out
run() { w=$1; shift
  timeout 200 python bench.py --workload $w --no-cpu-baseline "$@" 2>/dev/null | tail -1 > gpurun_out/bench_$w.json
  python - <<PY
import json
d=json.load(open("gpurun_out/bench_$w.json"))
o=d.get("one_launch_per_step") or {}
print("$w", "us/step %.3f"%(d["ms_per_step"]*1e3), "value %.3g"%d["value"], "GB/s %.0f"%d["roofline"]["achieved"], "frac %.3f"%d["roofline"]["frac"],
      "| one launch/step: us %.3f frac %.3f"%(o["roofline"]["launch_us_hip_events"], o["roofline"]["frac"]) if o else "")
PY
}
run hover65536_30hz
run hover65536_pid_240hz
run stack8x8192_ext_240hz
run multihover2x16384_240hz
run hover4m_240hz --steps 1024 --warmup 128
run hover16m_240hz --mode graph --steps 256 --warmup 64
run swarm65536_ext_240hz --steps 512 --warmup 64
