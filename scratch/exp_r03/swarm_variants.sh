#!/bin/bash
# kernel times of the swarm bench with variant builds of libgpd.so (what of dwg_force_kernel<2> costs what): scratch/exp/libgpd_<V>.so
for v in "" "$@"; do
  lib=""; [ -n "$v" ] && lib="GPD_LIB=$GRAFT_REPO_ROOT/scratch/exp/libgpd_$v.so"
  (cd /tmp && export TMPDIR=/tmp && env $lib timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_v -o p -- python $GRAFT_REPO_ROOT/bench.py --workload swarm65536_ext_240hz --steps 256 --warmup 64 --min-time 0.05 --no-cpu-baseline --no-parity > /dev/null 2>&1)
  f=$(find gpurun_out/prof_v -name "*kernel_stats.csv" | head -1)
  echo "== variant '${v:-shipped}'"; python - "$f" <<'PY'
import csv, sys
for i, r in enumerate(csv.DictReader(open(sys.argv[1]))):
    if i < 3: print("  ", r["Name"][27:60], r["Calls"], round(float(r["AverageNs"]) / 1e3, 2), "us")
PY
  rm -rf gpurun_out/prof_v
done
