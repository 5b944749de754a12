#!/bin/bash
# wave-local rollouts of every aviary size up to 64: the A/B table, then the whole GPU suite
O=gpurun_out/r06t; mkdir -p $O
timeout 600 python scratch/exp_r06/ab_wave_local.py > $O/ab_wave_local.log 2>&1; tail -21 $O/ab_wave_local.log
cp gpurun_out/r06s/ab_wave_local.json $O/
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; grep -n "passed\|failed" $O/pytest_gpu.log | tail -3; grep -n "^FAILED\|Error" $O/pytest_gpu.log | head
