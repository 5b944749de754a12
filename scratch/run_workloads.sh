for w in hover65536_30hz hover65536_pid_240hz stack8x8192_ext_240hz multihover2x16384_240hz hover4m_240hz; do
  timeout 150 python bench.py --workload $w --no-cpu-baseline $( [ $w = hover4m_240hz ] && echo "--steps 1024 --warmup 128" ) 2>/dev/null | tail -1 > gpurun_out/bench_$w.json
  python - <<PY
import json
d=json.load(open("gpurun_out/bench_$w.json"))
print("$w", "us/step %.3f"%(d["ms_per_step"]*1e3), "frac %.3f"%d["roofline"]["frac"], "2nd", (d.get("one_launch_per_step") or {}).get("ms_per_step"))
PY
done
