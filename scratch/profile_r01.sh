#!/bin/bash
# Collect the round-1 profiles: kernel-trace stats + HBM traffic counters (separate --pmc passes).
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/prof_r01
mkdir -p $OUT
for w in hover65536_240hz hover16m_240hz; do
  steps=1280; [ $w = hover16m_240hz ] && steps=256
  cd /tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$w -o t -- python $R/bench.py --workload $w --no-cpu-baseline --steps $steps --warmup 64 > $OUT/trace_$w.log 2>&1
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_${c}_$w -o p -- python $R/bench.py --workload $w --no-cpu-baseline --mode eager --steps 128 --warmup 16 > $OUT/pmc_${c}_$w.log 2>&1
  done
  cd $R
done
python - <<'PY'
import csv, glob, json, collections, os
out = {}
for w in ("hover65536_240hz", "hover16m_240hz"):
    rec = {}
    for f in glob.glob(f"gpurun_out/prof_r01/trace_{w}/**/*kernel_stats.csv", recursive=True):
        rows = list(csv.DictReader(open(f)))
        rec["kernel_stats"] = [r for r in rows if "gpd_" in r["Name"]][:4]
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        vals = []
        for f in glob.glob(f"gpurun_out/prof_r01/pmc_{c}_{w}/**/*counter_collection.csv", recursive=True):
            for row in csv.DictReader(open(f)):
                if "gpd_step_kernel" in row["Kernel_Name"] and row["Counter_Name"] == c:
                    vals.append(float(row["Counter_Value"]))
        vals = vals[len(vals)//4:]
        rec[c + "_KB_per_dispatch"] = sum(vals) / max(len(vals), 1)
        rec[c + "_n"] = len(vals)
    out[w] = rec
json.dump(out, open("gpurun_out/prof_r01/summary.json", "w"), indent=1)
print(json.dumps(out, indent=1)[:3000])
PY
