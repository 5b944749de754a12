#!/bin/bash
# A/B on ONE box, interleaved: the in-tree library against scratch/exp/libgpd_old.so on the workloads with the add-on force terms
run() { GPD_LIB=$1 timeout 200 python bench.py --workload $2 --no-cpu-baseline --no-second-leg --no-parity 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.load(sys.stdin); print('%-28s %-26s us/step %.4f  frac %.3f' % ('$2', '$1'[-26:], d['ms_per_step']*1e3, d['roofline']['frac']))"; }
for w in hover65536_ext_240hz stack8x8192_ext_240hz hover65536_240hz; do
  for rep in 1 2; do
    run gym-pybullet-drones_amd/csrc/libgpd.so $w
    run scratch/exp/libgpd_old.so $w
  done
done
