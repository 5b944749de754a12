#!/bin/bash
# A/B of library builds on ONE box: scratch/ab.sh <lib.so> ... (the default in-tree build first)
run() { GPD_LIB=$1 timeout 200 python bench.py --no-cpu-baseline --no-second-leg ${@:2} 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.load(sys.stdin); print('%-44s us/step %.4f  frac %.3f' % ('$1'[-44:], d['ms_per_step']*1e3, d['roofline']['frac']))"; }
for rep in 1 2; do
  run gym-pybullet-drones_amd/csrc/libgpd.so "${ARGS[@]}"
  for l in "$@"; do run $l; done
done
