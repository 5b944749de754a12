#!/bin/bash
# A/B of two libraries on the swarm lines (and the swarm tests on the new one).  usage: bash scratch/exp_r04/ab_swarm.sh tag
mkdir -p gpurun_out/r04a
for rep in 1 2; do
for lib in before after; do
  L="$PWD/gym-pybullet-drones_amd/csrc/libgpd.so"; [ $lib = before ] && L="$PWD/scratch/exp_r04/libgpd_before.so"
  GPD_LIB=$L python bench.py --workload swarm65536_ext_240hz --steps 240 --warmup 24 --no-cpu-baseline --no-parity 2>/dev/null | tail -1 > gpurun_out/r04a/ab_$1_${lib}_$rep.json
  python - gpurun_out/r04a/ab_$1_${lib}_$rep.json $lib <<'PY'
import json, sys
j = json.load(open(sys.argv[1])); print(sys.argv[2], "swarm65536 us/sub-step %.3f" % (j["ms_per_step"] * 1e3), "value %.4g" % j["value"])
PY
done; done
for lib in before after; do
  L="$PWD/gym-pybullet-drones_amd/csrc/libgpd.so"; [ $lib = before ] && L="$PWD/scratch/exp_r04/libgpd_before.so"
  GPD_LIB=$L python bench.py --workload swarm1m_ext_240hz --steps 64 --warmup 16 --no-cpu-baseline --no-parity 2>/dev/null | tail -1 > gpurun_out/r04a/ab_$1_${lib}_1m.json
  python - gpurun_out/r04a/ab_$1_${lib}_1m.json $lib <<'PY'
import json, sys
j = json.load(open(sys.argv[1])); print(sys.argv[2], "swarm1m us/sub-step %.3f" % (j["ms_per_step"] * 1e3), "value %.4g" % j["value"])
PY
done
python -m pytest tests/test_gpu_surface.py -q -k "swarm or world or stale or hipgraph or wake or halo" 2>&1 | tail -3
