#!/bin/bash
# usage: scratch/pmc.sh <outdir-name> <counters...>  (runs a short bench under rocprofv3 --pmc)
export TMPDIR=/tmp
R=$PWD
name=$1; shift
cd /tmp
rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/gpurun_out/pmc_$name -o pmc -- python $R/bench.py --no-cpu-baseline --mode eager --steps 200 --warmup 20 ${BENCH_ARGS} > $R/gpurun_out/pmc_$name.log 2>&1
cd $R
python - <<PY
import csv, glob, collections
files = glob.glob("gpurun_out/pmc_$name/**/*counter_collection.csv", recursive=True)
agg = collections.defaultdict(list)
for f in files:
    for row in csv.DictReader(open(f)):
        if "gpd_step_kernel" in row["Kernel_Name"]:
            agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, v in sorted(agg.items()):
    v = v[len(v)//4:]
    print("$name", k, "mean per dispatch %.4g" % (sum(v)/len(v)), "n", len(v))
PY
