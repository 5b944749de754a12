#!/bin/bash
# One GPU-box call of round 2: tests, the driver's bench command, the default command, experiments, profiles.
# usage: bash scratch/round2.sh [tests] [bench] [split] [workloads] [profile|profile-quick]
mkdir -p gpurun_out
for what in "$@"; do
case $what in
tests)
  timeout 1500 python -m pytest tests -m gpu -q -s > gpurun_out/pytest_r02.log 2>&1; echo "pytest rc $?"; grep -E "^(FAILED|ERROR)" gpurun_out/pytest_r02.log | head -20; tail -3 gpurun_out/pytest_r02.log | cut -c1-300 ;;
bench)
  timeout 300 python bench.py --steps 20 --warmup 5 2>gpurun_out/r02_bench_driver_cmd.err | tail -1 > gpurun_out/r02_bench_driver_cmd.json
  timeout 300 python bench.py --no-cpu-baseline 2>gpurun_out/r02_bench_default.err | tail -1 > gpurun_out/r02_bench_default.json
  python - <<'PY'
import json
for n in ("driver_cmd", "default"):
    try:
        d = json.load(open(f"gpurun_out/r02_bench_{n}.json"))
    except Exception as e:
        print(n, "NO LINE", e); continue
    o = d.get("one_launch_per_step") or {}
    print(n, "us/step %.4f" % (d["ms_per_step"] * 1e3), "value %.4g" % d["value"], "frac %.3f" % d["roofline"]["frac"], "repeats", d["repeats"],
          "wall us %.4f" % (d["wall_ms_per_step"] * 1e3), "| step: us %.3f frac %.3f" % (o["us_per_step"], o["roofline"]["frac"]) if o else "",
          "| clock", d.get("shader_clock_ghz_probe"), "| cpu", {k: (v.get("value"), v.get("cores")) for k, v in (d.get("cpu_baseline") or {}).items() if isinstance(v, dict)})
PY
  ;;
split)
  for c in 1 2 4 8; do
    timeout 200 python bench.py --mode graph --split $c --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r02_bench_split$c.json
    python -c "
import json; d=json.load(open('gpurun_out/r02_bench_split$c.json')); print('split $c', 'us/step %.3f'%(d['ms_per_step']*1e3), 'frac %.3f'%d['roofline']['frac'])"
  done ;;
debug-swarm)
  timeout 200 python scratch/debug_swarm.py eager 3000 > gpurun_out/debug_swarm_eager.log 2>&1; echo "eager rc $?"; tail -8 gpurun_out/debug_swarm_eager.log | cut -c1-250
  timeout 200 python scratch/debug_swarm.py graph 8192 > gpurun_out/debug_swarm_graph.log 2>&1; echo "graph rc $?"; tail -8 gpurun_out/debug_swarm_graph.log | cut -c1-250 ;;
floor)
  for k in 0 1; do HIP_FORCE_DEV_KERNARG=$k timeout 120 ./scratch/floor2; done > gpurun_out/r02_launch_floor_microbench.txt 2>&1; cat gpurun_out/r02_launch_floor_microbench.txt ;;
kernarg)
  for k in 0 1; do
    HIP_FORCE_DEV_KERNARG=$k timeout 200 python bench.py --mode graph --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r02_bench_kernarg$k.json
    python -c "
import json; d=json.load(open('gpurun_out/r02_bench_kernarg$k.json')); print('HIP_FORCE_DEV_KERNARG=$k graph', 'us/step %.3f'%(d['ms_per_step']*1e3), 'frac %.3f'%d['roofline']['frac'])"
  done ;;
workloads|workloads2|workloads3)
  if [ $what = workloads ]; then list="hover65536_30hz hover65536_pid_240hz stack8x8192_ext_240hz multihover2x16384_240hz hover65536_240hz_fullobs hover65536_30hz_fullobs hover65536_240hz_history hover65536_30hz_history swarm65536_ext_240hz hover4m_240hz"
  elif [ $what = workloads2 ]; then list="hover65536_240hz_fullobs hover65536_30hz_fullobs hover65536_240hz_history hover65536_30hz_history swarm65536_ext_240hz"
  else list="swarm65536_ext_240hz hover65536_30hz_policy hover65536_240hz_policy12 hover65536_30hz_policy_sample"; fi
  for w in $list; do
    timeout 300 python -X faulthandler bench.py --workload $w --no-cpu-baseline > gpurun_out/r02_bench_$w.out 2>gpurun_out/r02_bench_$w.err; echo "$w rc $?"
    tail -1 gpurun_out/r02_bench_$w.out > gpurun_out/r02_bench_$w.json
    python - <<PY
import json
try:
    d=json.load(open("gpurun_out/r02_bench_$w.json"))
    o=d.get("one_launch_per_step") or {}
    print("$w", "us/step %.3f"%(d["ms_per_step"]*1e3), "value %.3g"%d["value"], "frac %.3f"%d["roofline"]["frac"],
          "| one launch/step: us %.3f frac %.3f"%(o["us_per_step"], o["roofline"]["frac"]) if o else "")
except Exception as e:
    print("$w", "FAILED", e, open("gpurun_out/r02_bench_$w.err").read()[-600:])
PY
  done ;;
profile) timeout 1500 python scratch/profile_r02.py > gpurun_out/profile_r02.log 2>&1; tail -40 gpurun_out/profile_r02.log | cut -c1-400 ;;
profile-quick) timeout 900 python scratch/profile_r02.py quick > gpurun_out/profile_r02.log 2>&1; tail -30 gpurun_out/profile_r02.log | cut -c1-400 ;;
esac
done
