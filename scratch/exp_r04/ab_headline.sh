#!/bin/bash
# A/B of two libraries on the headline command (K = 20, the driver's) and the default one (K = 4096 -> 64 steps per launch).
mkdir -p gpurun_out/r04a
show() { python - "$1" "$2" <<'PY'
import json, sys
j = json.load(open(sys.argv[1])); o = j.get("one_launch_per_step") or {}
print(sys.argv[2], "us/step %.4f" % (j["ms_per_step"] * 1e3), "frac %.3f" % j["roofline"]["frac"], "parity", j.get("parity", {}).get("max"), "| step leg us %.3f" % o.get("us_per_step", 0))
PY
}
for rep in 1 2 3; do
for lib in before after; do
  L="gym-pybullet-drones_amd/csrc/libgpd.so"; [ $lib = before ] && L="scratch/exp_r04/libgpd_before.so"
  GPD_LIB=$L python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-hbm-leg 2>/dev/null | tail -1 > gpurun_out/r04a/ab_$1_${lib}_k20_$rep.json; show gpurun_out/r04a/ab_$1_${lib}_k20_$rep.json "$lib K=20"
done; done
for lib in before after; do
  L="gym-pybullet-drones_amd/csrc/libgpd.so"; [ $lib = before ] && L="scratch/exp_r04/libgpd_before.so"
  GPD_LIB=$L python bench.py --no-cpu-baseline --no-hbm-leg 2>/dev/null | tail -1 > gpurun_out/r04a/ab_$1_${lib}_k64.json; show gpurun_out/r04a/ab_$1_${lib}_k64.json "$lib K=4096"
  for wl in hover65536_30hz hover65536_pid_240hz stack8x8192_ext_240hz hover4m_240hz; do
    GPD_LIB=$L python bench.py --workload $wl --steps 20 --warmup 5 --no-cpu-baseline --no-second-leg --no-parity 2>/dev/null | tail -1 > gpurun_out/r04a/ab_$1_${lib}_$wl.json; show gpurun_out/r04a/ab_$1_${lib}_$wl.json "$lib $wl K=20"
  done
done
