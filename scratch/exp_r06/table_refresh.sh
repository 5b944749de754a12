#!/bin/bash
# DESIGN.md section 5's table on the final library: every rollout row once (64 steps per launch) with its one-launch-per-step leg
O=gpurun_out/r06t; mkdir -p $O
for w in hover65536_240hz hover65536_30hz hover65536_pid_240hz hover65536_ext_240hz hover4096_240hz stack8x8192_ext_240hz multihover2x16384_240hz hover65536_240hz_termobs hover65536_ext_pid_240hz stack8x8192_ext_pid_240hz multihover2x16384_pid_240hz hover4096_30hz hover4096_pid_240hz; do
  python bench.py --workload $w --steps 64 --warmup 64 --no-cpu-baseline --no-hbm-leg --no-dropin-leg --min-time 0.5 > $O/$w.json 2>/dev/null
  python - "$O/$w.json" "$w" <<'P'
import json,sys
j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
o=j.get('one_launch_per_step',{})
print(f"{sys.argv[2]:32s} rollout {j['ms_per_step']*1e3:7.4f} us/step  {j['value']:.3e}  frac {j['roofline']['frac']:.3f}  one launch per step {o.get('us_per_step',float('nan')):6.3f}  parity {j['parity']['ok']} {j['parity']['max']:.2e}")
P
done
