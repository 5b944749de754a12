#!/bin/bash
run() { GPD_LIB=$1 timeout 300 python bench.py --no-cpu-baseline --no-second-leg ${@:2} 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.load(sys.stdin); print('%-40s %-22s us/step %.4f  frac %.3f' % ('$1'[-40:], d['config']['workload'], d['ms_per_step']*1e3, d['roofline']['frac']))"; }
for rep in 1 2; do
for w in hover4m_240hz hover65536_30hz multihover2x16384_240hz hover4096_240hz; do
  run gym-pybullet-drones_amd/csrc/libgpd.so --workload $w
  run scratch/exp/libgpd_PLAIN_OBS_STORES.so --workload $w
done; done
