#!/bin/bash
# gpd_pid_sync + the factored completion word: the controller / host-visible / fixture tests, the A/B again
O=gpurun_out/r06n; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q -k "step_sync or host_visible or dropin or dslpid or pid_circle or c_host or examples" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout 600 python scratch/exp_r06/ab_step_sync.py > $O/ab.log 2>&1; tail -1 $O/ab.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(json.dumps(d['median_of_4_rounds'], indent=1))"
