#!/bin/bash
# Round-5 bench lines with the final library (GPU box): gpurun_out/bench_r05/<name>.json; scratch/refresh_profiles_r05.py bench copies them to profiles/r05_bench_*.json
cd "$(dirname "$0")/.."
O=gpurun_out/bench_r05
mkdir -p $O
python bench.py --steps 20 --warmup 5 > $O/driver_cmd.json 2> $O/driver_cmd.err
python bench.py > $O/default.json 2> $O/default.err
for w in hover65536_30hz hover65536_pid_240hz hover65536_ext_240hz hover65536_ext_pid_240hz hover4096_240hz hover4096_30hz hover4096_pid_240hz \
         stack8x8192_ext_240hz stack8x8192_ext_pid_240hz multihover2x16384_240hz multihover2x16384_pid_240hz hover65536_240hz_termobs \
         hover65536_240hz_history hover65536_30hz_history hover4m_240hz; do
    python bench.py --workload $w --no-cpu-baseline > $O/$w.json 2> $O/$w.err
done
python bench.py --workload swarm65536_ext_240hz --steps 240 --warmup 24 > $O/swarm65536_ext_240hz.json 2> $O/swarm65536.err
python bench.py --workload swarm1m_ext_240hz --steps 64 --warmup 16 --no-cpu-baseline > $O/swarm1m_ext_240hz.json 2> $O/swarm1m.err
GPD_DIST_BACKEND=gloo GPD_BENCH_SINGLE_DEVICE=1 python bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline > $O/two_ranks_one_device.json 2> $O/two_ranks.err
for f in $O/*.json; do python - "$f" <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    o = j.get("one_launch_per_step") or {}
    print("%-40s %8.3f us/step frac %.3f | graph %s us | parity %s" % (sys.argv[1].split("/")[-1], j["ms_per_step"] * 1e3, j["roofline"]["frac"], ("%.3f" % o["us_per_step"]) if o else "-", (j.get("parity") or {}).get("max")))
except Exception as e:
    print(sys.argv[1], "UNREADABLE", e)
PY
done
