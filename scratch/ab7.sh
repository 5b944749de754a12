#!/bin/bash
run() { GPD_LIB=$1 timeout 200 python bench.py --no-cpu-baseline --workload $2 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.load(sys.stdin); o=d.get('one_launch_per_step') or {}; print('%-34s %-26s rollout us %.4f  step us %.4f' % ('$1'[-34:], '$2', d['ms_per_step']*1e3, o.get('us_per_step', 0)))"; }
for w in hover65536_240hz hover65536_30hz hover65536_pid_240hz hover65536_30hz_policy; do
  for l in gym-pybullet-drones_amd/csrc/libgpd.so "$@"; do run $l $w; done; done
