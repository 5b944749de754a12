#!/bin/bash
# usage: scratch/ab.sh "<workloads>" <lib1> <lib2> ...   -> us/launch for each lib x workload, 2 rounds interleaved
WL=$1; shift
for round in 1 2; do for lib in "$@"; do for w in $WL; do
  GPD_LIB=$lib python bench.py --workload $w --no-cpu-baseline --steps 1920 --warmup 192 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$lib', d['config']['workload'], 'us/launch %.2f'%d['roofline']['launch_us_hip_events'], 'GB/s %.0f'%d['roofline']['achieved'])
"; done; done; done
