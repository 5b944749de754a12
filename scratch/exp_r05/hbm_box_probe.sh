#!/bin/bash
# What kind of box is this?  The hover4m rollout leg with non-temporal / plain observation stores, and the copy / fill probes, three rounds.
cd "$(dirname "$0")/../.."
for r in 1 2 3; do
  for mode in nt plain; do
    if [ $mode = plain ]; then export GPD_ROLLOUT_OBS_STORES=plain; else unset GPD_ROLLOUT_OBS_STORES; fi
    python bench.py --workload hover4m_240hz --steps 64 --warmup 64 --no-cpu-baseline --no-second-leg --no-parity --min-time 1.0 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('round $r obs stores $mode: %.1f us per launch, frac %.3f' % (j['roofline']['launch_us_hip_events'], j['roofline']['frac']))"
  done
done
python - <<'PY'
import sys, torch
sys.path.insert(0, '.')
import bench
print(bench.copy_probe(torch.device('cuda:0'), mib=1024, reps=100))
PY
