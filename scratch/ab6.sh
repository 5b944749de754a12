#!/bin/bash
run() { GPD_LIB=$1 timeout 200 python bench.py --no-cpu-baseline --no-second-leg --workload $2 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.load(sys.stdin); print('%-36s %-26s us/step %.4f' % ('$1'[-36:], '$2', d['ms_per_step']*1e3))"; }
for w in hover65536_30hz_policy hover65536_240hz_policy12; do
  for l in gym-pybullet-drones_amd/csrc/libgpd.so "$@"; do run $l $w; done; done
