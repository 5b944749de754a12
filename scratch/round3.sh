#!/bin/bash
# One GPU-box call of round 3.  usage: bash scratch/round3.sh [diag] [tests] [bench] [split] [configs] [swarm] ...
mkdir -p gpurun_out
show() { python - "$1" "$2" <<'PY'
import json, sys
name, path = sys.argv[1], sys.argv[2]
try:
    d = json.load(open(path))
except Exception as e:
    print(name, "NO LINE", e); sys.exit(0)
o = d.get("one_launch_per_step") or {}
p = d.get("parity") or {}
print(name, "us/step %.4f" % (d["ms_per_step"] * 1e3), "value %.4g" % d["value"], "frac %.3f" % d["roofline"]["frac"], "B/drone/step %.1f" % d["roofline"]["bytes_per_drone_per_env_step"],
      "repeats", d["repeats"], "wall us %.4f" % (d["wall_ms_per_step"] * 1e3), "chains", d["roofline"].get("concurrent_chains"),
      ("| step: us %.3f frac %.3f" % (o["us_per_step"], o["roofline"]["frac"])) if o else "",
      "| parity", {k: (("%.2e" % v) if isinstance(v, float) else v) for k, v in p.items() if k in ("checked_steps", "pos", "quat", "vel", "rates", "flag_mismatch_frac", "reward_max_abs", "max", "ok", "error", "episodes_ended_in_window")})
PY
}
for what in "$@"; do
case $what in
diag)
  for f in 15 31; do timeout 300 python scratch/smoke_diag.py $f; done > gpurun_out/r03_smoke_diag.txt 2>&1; echo "diag rc $?"; cat gpurun_out/r03_smoke_diag.txt | cut -c1-330
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r03_smoke.log 2>&1; echo "smoke rc $?"; tail -12 gpurun_out/r03_smoke.log | cut -c1-400 ;;
tests)
  timeout 1800 python -m pytest tests -m gpu -q -s -x > gpurun_out/pytest_r03.log 2>&1; echo "pytest rc $?"; grep -E "^(FAILED|ERROR)" gpurun_out/pytest_r03.log | head -20; tail -5 gpurun_out/pytest_r03.log | cut -c1-600 ;;
tests-all)
  timeout 1800 python -m pytest tests -m gpu -q > gpurun_out/pytest_r03.log 2>&1; echo "pytest rc $?"; grep -E "^(FAILED|ERROR)" gpurun_out/pytest_r03.log | head -30; tail -5 gpurun_out/pytest_r03.log | cut -c1-600 ;;
newtests)
  timeout 900 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -s -k "timed_workload" > gpurun_out/pytest_r03_new.log 2>&1; echo "pytest rc $?"; grep -E "^(FAILED|ERROR)|checked_steps" gpurun_out/pytest_r03_new.log | cut -c1-900 | head; tail -3 gpurun_out/pytest_r03_new.log | cut -c1-300
  timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_rollout.py -m gpu -q -k "16 or 24 or 31" > gpurun_out/pytest_r03_damp.log 2>&1; echo "pytest damp rc $?"; grep -E "^(FAILED|ERROR)" gpurun_out/pytest_r03_damp.log | head; tail -3 gpurun_out/pytest_r03_damp.log | cut -c1-300 ;;
bench)
  timeout 400 python bench.py --steps 20 --warmup 5 2>gpurun_out/r03_bench_driver_cmd.err | tail -1 > gpurun_out/r03_bench_driver_cmd.json; show driver_cmd gpurun_out/r03_bench_driver_cmd.json
  timeout 400 python bench.py --no-cpu-baseline 2>gpurun_out/r03_bench_default.err | tail -1 > gpurun_out/r03_bench_default.json; show default gpurun_out/r03_bench_default.json ;;
split)
  for k in 20 64; do for c in 1 2 4 8; do
    timeout 200 python bench.py --steps $k --warmup 5 --split $c --no-cpu-baseline --no-second-leg 2>gpurun_out/r03_bench_rollout${k}_split$c.err | tail -1 > gpurun_out/r03_bench_rollout${k}_split$c.json
    show "rollout$k split $c" gpurun_out/r03_bench_rollout${k}_split$c.json
  done; done ;;
splitgraph)
  for k in 20 64; do for c in 1 2 4; do
    timeout 200 python bench.py --steps $k --warmup 5 --split $c --rollout-graph 8 --no-parity --no-cpu-baseline --no-second-leg 2>gpurun_out/r03_bench_rollout${k}_split${c}_graph.err | tail -1 > gpurun_out/r03_bench_rollout${k}_split${c}_graph.json
    show "rollout$k split $c, 8 passes per hipGraph" gpurun_out/r03_bench_rollout${k}_split${c}_graph.json
  done; done ;;
mask2)
  for k in 20 64; do for c in 2 4; do
    timeout 200 python bench.py --steps $k --warmup 5 --split $c --cu-mask --no-parity --no-cpu-baseline --no-second-leg 2>gpurun_out/r03_bench_rollout${k}_split${c}_cumask2.err | tail -1 > gpurun_out/r03_bench_rollout${k}_split${c}_cumask2.json
    show "rollout$k split $c cu-mask (blocked bits)" gpurun_out/r03_bench_rollout${k}_split${c}_cumask2.json
  done; done ;;
mask)
  for k in 20 64; do for c in 2 4 8; do
    timeout 200 python bench.py --steps $k --warmup 5 --split $c --cu-mask --no-cpu-baseline --no-second-leg 2>gpurun_out/r03_bench_rollout${k}_split${c}_cumask.err | tail -1 > gpurun_out/r03_bench_rollout${k}_split${c}_cumask.json
    show "rollout$k split $c cu-mask" gpurun_out/r03_bench_rollout${k}_split${c}_cumask.json
  done; done ;;
swarmtests)
  timeout 900 python -m pytest tests/test_gpu_surface.py -m gpu -q -x -k "swarm or world or stale or hipgraph or wake" > gpurun_out/pytest_r03_swarm.log 2>&1; echo "pytest swarm rc $?"; grep -E "^(FAILED|ERROR)|Error|assert" gpurun_out/pytest_r03_swarm.log | head -20; tail -3 gpurun_out/pytest_r03_swarm.log | cut -c1-300 ;;
swarm)
  for cfg in "10.0 1" "10.5 4" "10.5 8" "10.5 16" "11.0 16"; do set -- $cfg
    GPD_SWARM_CELL=$1 GPD_SWARM_REBIN=$2 timeout 300 python bench.py --workload swarm65536_ext_240hz --steps 240 --warmup 24 --no-cpu-baseline 2>gpurun_out/r03_swarm_c$1_m$2.err | tail -1 > gpurun_out/r03_swarm_c$1_m$2.json
    show "swarm65536 cell $1 rebin $2" gpurun_out/r03_swarm_c$1_m$2.json
  done ;;
swarm1mdbg)
  timeout 300 python scratch/debug_swarm1m.py > gpurun_out/r03_debug_swarm1m.txt 2>&1; echo "rc $?"; grep -v Warning gpurun_out/r03_debug_swarm1m.txt | cut -c1-200 | tail -40 ;;
partition)
  timeout 600 python scratch/swarm_partition_cost.py > gpurun_out/r03_swarm_partition_cost.txt 2>&1; echo "rc $?"; grep -v "Warning\|SwarmAviary(\|amdgpu.ids" gpurun_out/r03_swarm_partition_cost.txt | cut -c1-200 ;;
swarmdbg)
  timeout 300 python scratch/debug_swarm3.py > gpurun_out/r03_debug_swarm3.txt 2>&1; echo "rc $?"; grep -v Warning gpurun_out/r03_debug_swarm3.txt | cut -c1-260 ;;
swarmprof)
  for cfg in "10.0 1" "10.5 8"; do set -- $cfg
    (cd /tmp && export TMPDIR=/tmp && GPD_SWARM_CELL=$1 GPD_SWARM_REBIN=$2 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_swarm_c$1_m$2 -o p -- python $GRAFT_REPO_ROOT/bench.py --workload swarm65536_ext_240hz --steps 256 --warmup 64 --min-time 0.05 --no-cpu-baseline > /dev/null 2>&1)
    f=$(find gpurun_out/prof_swarm_c$1_m$2 -name "*kernel_stats.csv" | head -1); echo "== cell $1 rebin $2: $f"; head -12 $f | cut -c1-200
  done ;;
swarmq)   # the swarm line + the kernel times of the same command (default cell / rebin)
  timeout 300 python bench.py --workload swarm65536_ext_240hz --steps 240 --warmup 24 --no-cpu-baseline 2>gpurun_out/r03_swarmq.err | tail -1 > gpurun_out/r03_swarmq.json
  show "swarm65536" gpurun_out/r03_swarmq.json
  (cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_swarmq -o p -- python $GRAFT_REPO_ROOT/bench.py --workload swarm65536_ext_240hz --steps 256 --warmup 64 --min-time 0.05 --no-cpu-baseline --no-parity > /dev/null 2>&1)
  f=$(find gpurun_out/prof_swarmq -name "*kernel_stats.csv" | head -1); head -8 $f | cut -c1-160; cp $f gpurun_out/r03_swarmq_kernel_stats.csv; rm -rf gpurun_out/prof_swarmq ;;
swarm2)
  for cfg in "10.25 8" "10.25 16" "10.5 16" "10.5 32" "10.75 32" "11.0 32"; do set -- $cfg
    GPD_SWARM_CELL=$1 GPD_SWARM_REBIN=$2 timeout 300 python bench.py --workload swarm65536_ext_240hz --steps 240 --warmup 24 --no-cpu-baseline 2>gpurun_out/r03_swarm_c$1_m$2.err | tail -1 > gpurun_out/r03_swarm_c$1_m$2.json
    show "swarm65536 cell $1 rebin $2" gpurun_out/r03_swarm_c$1_m$2.json
  done ;;
regress)
  for w in hover65536_30hz hover65536_pid_240hz stack8x8192_ext_240hz multihover2x16384_240hz hover4m_240hz; do
    extra=""; case $w in hover4m*) extra="--steps 256 --warmup 64" ;; esac
    timeout 400 python bench.py --workload $w $extra --no-cpu-baseline 2>gpurun_out/r03_bench_$w.err | tail -1 > gpurun_out/r03_bench_$w.json; show $w gpurun_out/r03_bench_$w.json
  done ;;
swarm1m)
  timeout 600 python bench.py --workload swarm1m_ext_240hz --steps 256 --warmup 16 --no-cpu-baseline 2>gpurun_out/r03_bench_swarm1m.err | tail -1 > gpurun_out/r03_bench_swarm1m_ext_240hz.json; show swarm1m gpurun_out/r03_bench_swarm1m_ext_240hz.json ;;
profile) timeout 1500 python scratch/profile_r03.py quick > gpurun_out/profile_r03.log 2>&1; tail -30 gpurun_out/profile_r03.log | cut -c1-300 ;;
profile-full) timeout 2400 python scratch/profile_r03.py > gpurun_out/profile_r03.log 2>&1; tail -40 gpurun_out/profile_r03.log | cut -c1-300 ;;
configs)
  for w in hover4096_240hz hover65536_ext_240hz hover65536_240hz_termobs hover65536_240hz_history hover65536_30hz_history swarm65536_ext_240hz; do
    extra=""; case $w in swarm*) extra="--steps 240 --warmup 24" ;; esac
    timeout 400 python bench.py --workload $w $extra 2>gpurun_out/r03_bench_$w.err | tail -1 > gpurun_out/r03_bench_$w.json; show $w gpurun_out/r03_bench_$w.json
  done ;;
configs2)   # config 2 at 30 Hz and the closed-loop run of every config
  for w in hover4096_30hz hover4096_pid_240hz hover65536_ext_pid_240hz stack8x8192_ext_pid_240hz multihover2x16384_pid_240hz; do
    timeout 400 python bench.py --workload $w 2>gpurun_out/r03_bench_$w.err | tail -1 > gpurun_out/r03_bench_$w.json; show $w gpurun_out/r03_bench_$w.json
  done ;;
*) echo "unknown stage $what" ;;
esac
done
