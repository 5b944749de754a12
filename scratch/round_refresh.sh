#!/bin/bash
# One GPU-box call that regenerates everything profiles/ holds for the round (then: python scratch/refresh_profiles.py here)
mkdir -p gpurun_out
timeout 300 python bench.py 2>/dev/null | tail -1 > gpurun_out/bench_default.json
python -c "
import json; d=json.load(open('gpurun_out/bench_default.json')); print('default', d['ms_per_step']*1e3, d['value'], d['roofline']['frac'], d['cpu_baseline']['value'])"
bash scratch/run_workloads.sh
timeout 900 bash scratch/profile_r01.sh > gpurun_out/profile_r01.log 2>&1; tail -3 gpurun_out/profile_r01.log | cut -c1-300
timeout 100 bash scratch/pmc_rollout.sh sq1 hover65536_240hz SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES > gpurun_out/sq_counters.txt 2>&1
timeout 100 bash scratch/pmc_rollout.sh sq2 hover65536_240hz SQ_WAVE_CYCLES SQ_INSTS_BRANCH SQ_IFETCH SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_WAIT_INST_LDS >> gpurun_out/sq_counters.txt 2>&1
cat gpurun_out/sq_counters.txt | grep "^sq"
[ -x scratch/issue ] && timeout 30 ./scratch/issue > gpurun_out/issue_microbench.txt 2>&1
