#!/bin/bash
# A/B of two libraries: the one-world lines and the one-launch-per-step leg of the add-on-force workloads (the last-RPM loads of
# the drag term issued with the rest of the state instead of behind the first wait), then the GPU tests that cover those kernels.
mkdir -p gpurun_out/r04a
line() { python - "$@" <<'PY'
import json, sys
j = json.load(open(sys.argv[1])); o = j.get("one_launch_per_step") or {}
print(sys.argv[2], sys.argv[3], "us/step %.3f" % (j["ms_per_step"] * 1e3), ("| one launch per step %.3f us" % o["us_per_step"]) if o else "")
PY
}
for rep in 1 2; do for lib in before after; do
  L="$PWD/gym-pybullet-drones_amd/csrc/libgpd.so"; [ $lib = before ] && L="$PWD/scratch/exp_r04/libgpd_before.so"
  f=gpurun_out/r04a/ab_$1_${lib}_swarm65536_$rep.json
  GPD_LIB=$L python bench.py --workload swarm65536_ext_240hz --steps 240 --warmup 24 --no-cpu-baseline --no-parity 2>/dev/null | tail -1 > $f; line $f $lib swarm65536
  f=gpurun_out/r04a/ab_$1_${lib}_ext_$rep.json
  GPD_LIB=$L python bench.py --workload hover65536_ext_240hz --steps 64 --warmup 64 --no-cpu-baseline --no-parity --min-time 0.1 2>/dev/null | tail -1 > $f; line $f $lib hover65536_ext
done; done
for lib in before after; do
  L="$PWD/gym-pybullet-drones_amd/csrc/libgpd.so"; [ $lib = before ] && L="$PWD/scratch/exp_r04/libgpd_before.so"
  f=gpurun_out/r04a/ab_$1_${lib}_1m.json
  GPD_LIB=$L python bench.py --workload swarm1m_ext_240hz --steps 64 --warmup 16 --no-cpu-baseline --no-parity 2>/dev/null | tail -1 > $f; line $f $lib swarm1m
done
python -m pytest tests/test_gpu_parity.py -q -x 2>&1 | tail -2; python -m pytest tests/test_gpu_surface.py -q -k "swarm or world or stale or hipgraph or wake or halo" 2>&1 | tail -2
