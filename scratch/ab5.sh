#!/bin/bash
run() { GPD_LIB=$1 timeout 200 python bench.py --no-cpu-baseline --mode graph --workload $2 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.load(sys.stdin); print('%-36s %-26s step us %.4f' % ('$1'[-36:], '$2', d['ms_per_step']*1e3))"; }
for w in hover65536_240hz hover65536_30hz hover65536_pid_240hz; do
  for rep in 1 2; do run scratch/exp/libgpd_BEFORE.so $w; run gym-pybullet-drones_amd/csrc/libgpd.so $w; done; done
