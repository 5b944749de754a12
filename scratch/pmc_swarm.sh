#!/bin/bash
# Round 3: SQ / HBM counters of the swarm kernels (eager launches so that every dispatch is a traced dispatch)
mkdir -p gpurun_out/pmc_swarm
cd /tmp && export TMPDIR=/tmp
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU" "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_swarm/$tag -o p -- python $GRAFT_REPO_ROOT/bench.py --workload swarm65536_ext_240hz --mode eager --steps 96 --warmup 16 --min-time 0.001 --no-cpu-baseline --no-parity > /dev/null 2>&1
  echo "$tag rc $?"
done
cd $GRAFT_REPO_ROOT
python - <<'PY'
import collections, csv, glob
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_swarm/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        name = "force<2> replay" if "dwg_force_kernel<2>" in k else "force<1> build" if "dwg_force_kernel<1>" in k else "swarm_step" if "gpd_swarm_step" in k else None
        if name:
            agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = []
for name, cs in agg.items():
    line = [name] + [f"{c}={sum(v[len(v)//4:]) / max(1, len(v) - len(v)//4):.4g}" for c, v in sorted(cs.items())]
    out.append("  ".join(line))
open("gpurun_out/r03_swarm_counters.txt", "w").write("\n".join(out) + "\n")
print("\n".join(out))
PY
