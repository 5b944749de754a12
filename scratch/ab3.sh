#!/bin/bash
run() { GPD_LIB=$1 timeout 200 python bench.py --no-cpu-baseline --mode graph 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.load(sys.stdin); print('%-44s step us %.4f  frac %.3f' % ('$1'[-44:], d['ms_per_step']*1e3, d['roofline']['frac']))"; }
for rep in 1 2; do run gym-pybullet-drones_amd/csrc/libgpd.so; for l in "$@"; do run $l; done; done
