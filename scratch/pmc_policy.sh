#!/bin/bash
# SQ counters of the policy rollout kernel (two passes), N = 65536
export TMPDIR=/tmp
R=$PWD
for w in hover65536_30hz_policy hover65536_240hz_policy12; do
for g in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES"; do
  tag=$(echo $g | cut -d' ' -f2)
  (cd /tmp && rocprofv3 --kernel-trace --pmc $g --output-format csv -d $R/gpurun_out/pmcpol_${w}_$tag -o p -- python $R/bench.py --workload $w --no-cpu-baseline --no-second-leg --steps 64 --warmup 64 --min-time 0.002 > $R/gpurun_out/pmcpol_${w}_$tag.log 2>&1)
  python - <<PY
import csv, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob("gpurun_out/pmcpol_${w}_$tag/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "policy" in row["Kernel_Name"]:
            agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, v in sorted(agg.items()):
    v = v[len(v)//4:]
    print("$w", k, "per wave-step %.1f" % (sum(v)/len(v)/1024/64), "n", len(v))
PY
done; done
