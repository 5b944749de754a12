#!/bin/bash
# usage: scratch/pmc_rollout.sh <tag> <workload> <counters...>   (rollout mode, few launches)
export TMPDIR=/tmp
R=$PWD
tag=$1; w=$2; shift; shift
cd /tmp
rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/gpurun_out/pmcr_$tag -o p -- python $R/bench.py --workload $w --no-cpu-baseline --no-second-leg --steps 256 --warmup 64 > $R/gpurun_out/pmcr_$tag.log 2>&1
cd $R
python - <<PY
import csv, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob("gpurun_out/pmcr_$tag/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "gpd_step_kernel" in row["Kernel_Name"] or "gpd_rollout" in row["Kernel_Name"]:
            agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, v in sorted(agg.items()):
    print("$tag", k, "mean per dispatch %.6g" % (sum(v)/len(v)), "n", len(v))
PY
