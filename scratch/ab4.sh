#!/bin/bash
run() { GPD_LIB=$1 timeout 200 python bench.py --no-cpu-baseline --mode graph --workload $2 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.load(sys.stdin); print('%-36s %-26s step us %.4f' % ('$1'[-36:], '$2', d['ms_per_step']*1e3))"; }
for w in hover4096_240hz hover65536_30hz hover65536_pid_240hz multihover2x16384_240hz stack8x8192_ext_240hz; do
  for rep in 1 2; do run gym-pybullet-drones_amd/csrc/libgpd.so $w; run scratch/exp/libgpd_STEPBIG.so $w; done; done
