#!/bin/bash
# Round-1 profiles: rocprofv3 kernel-trace stats of the default bench command (rollout headline + one-launch-per-step
# leg) and HBM traffic counters in separate --pmc passes (FETCH_SIZE / WRITE_SIZE), for the headline workload and a
# large batch.  Output: gpurun_out/prof_r01/ (+ summary.json); copy what is to be judged into profiles/.
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/prof_r01
rm -rf $OUT; mkdir -p $OUT
cd /tmp
# (1) the exact default bench command under --kernel-trace --stats
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_default -o t -- python $R/bench.py --no-cpu-baseline > $OUT/trace_default.log 2>&1
# (2) large batch, both launch modes
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_hover4m -o t -- python $R/bench.py --workload hover4m_240hz --no-cpu-baseline --steps 512 --warmup 64 > $OUT/trace_hover4m.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_hover16m -o t -- python $R/bench.py --workload hover16m_240hz --mode graph --no-cpu-baseline --steps 256 --warmup 64 > $OUT/trace_hover16m.log 2>&1
# (3) HBM traffic, separate passes per counter and per launch mode
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_${c}_hover65536_rollout -o p -- python $R/bench.py --no-cpu-baseline --no-second-leg --steps 512 --warmup 64 > $OUT/pmc_${c}_hover65536_rollout.log 2>&1
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_${c}_hover65536_step -o p -- python $R/bench.py --no-cpu-baseline --mode eager --steps 128 --warmup 16 > $OUT/pmc_${c}_hover65536_step.log 2>&1
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_${c}_hover4m_rollout -o p -- python $R/bench.py --workload hover4m_240hz --no-cpu-baseline --no-second-leg --steps 128 --warmup 64 > $OUT/pmc_${c}_hover4m_rollout.log 2>&1
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_${c}_hover16m_step -o p -- python $R/bench.py --workload hover16m_240hz --no-cpu-baseline --mode eager --steps 64 --warmup 16 > $OUT/pmc_${c}_hover16m_step.log 2>&1
done
cd $R
python - <<'PY'
import csv, glob, json
out = {}
def stats(tag):
    for f in glob.glob(f"gpurun_out/prof_r01/{tag}/**/*kernel_stats.csv", recursive=True):
        return [r for r in csv.DictReader(open(f)) if "gpd_" in r["Name"]]
    return []
for tag in ("trace_default", "trace_hover4m", "trace_hover16m"):
    out[tag] = stats(tag)
for run, kern in (("hover65536_rollout", "gpd_rollout"), ("hover65536_step", "gpd_step_kernel"),
                  ("hover4m_rollout", "gpd_rollout"), ("hover16m_step", "gpd_step_kernel")):
    rec = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        vals = []
        for f in glob.glob(f"gpurun_out/prof_r01/pmc_{c}_{run}/**/*counter_collection.csv", recursive=True):
            for row in csv.DictReader(open(f)):
                if kern in row["Kernel_Name"] and row["Counter_Name"] == c:
                    vals.append(float(row["Counter_Value"]))
        vals = vals[len(vals) // 4:]
        rec[c + "_KB_per_dispatch"] = sum(vals) / max(len(vals), 1)
        rec[c + "_n"] = len(vals)
    out["pmc_" + run] = rec
json.dump(out, open("gpurun_out/prof_r01/summary.json", "w"), indent=1)
print(json.dumps(out, indent=1)[:6000])
PY
for d in $OUT/trace_*; do n=$(basename $d); f=$(find $d -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/${n}_kernel_stats.csv; done
for d in $OUT/pmc_*; do [ -d $d ] || continue; n=$(basename $d); f=$(find $d -name "*counter_collection.csv" | head -1); [ -n "$f" ] && grep -E "Counter_Name|gpd_" $f | head -400 > $OUT/${n}.csv; done
tail -2 $OUT/trace_default.log | head -1 | cut -c1-1500
