#!/bin/bash
# the completion word of gpd_step_sync: its tests, the drop-in latency, the driver's command
O=gpurun_out/r06l; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_surface.py -m gpu -x -q -k "step_sync or host_visible or dropin" > $O/pytest_sync.log 2>&1; tail -3 $O/pytest_sync.log
timeout 300 python scratch/exp_r06/dropin_latency.py > $O/dropin_latency.log 2>&1; grep host_visible $O/dropin_latency.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-hbm-leg > $O/bench_driver_nohbm.json 2> $O/bench.err
python - <<'P'
import json
p=json.loads(open('gpurun_out/r06l/bench_driver_nohbm.json').read().strip().splitlines()[-1])
print(p['value'], p['roofline']['frac'], p['one_launch_per_step']['us_per_step'], p['python_step_loop']['us_per_step'], p['parity']['ok'], p['dropin_single_env']['us_per_step'], p['dropin_single_env']['breakdown_us'])
P
