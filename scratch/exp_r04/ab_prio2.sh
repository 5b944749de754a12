#!/bin/bash
# thresholds of the replay waves' priority: before = no priority, prio1 = 12/16/20, after = 11/13/16
for lib in before prio1 after before prio1 after; do
  L="$PWD/gym-pybullet-drones_amd/csrc/libgpd.so"; [ $lib = before ] && L="$PWD/scratch/exp_r04/libgpd_before.so"; [ $lib = prio1 ] && L="$PWD/scratch/exp_r04/libgpd_prio1.so"
  GPD_LIB=$L python bench.py --workload swarm65536_ext_240hz --steps 240 --warmup 24 --no-cpu-baseline --no-parity 2>/dev/null | tail -1 | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('$lib swarm65536 us/sub-step %.3f' % (j['ms_per_step']*1e3))"
done
for lib in before prio1 after before after; do
  L="$PWD/gym-pybullet-drones_amd/csrc/libgpd.so"; [ $lib = before ] && L="$PWD/scratch/exp_r04/libgpd_before.so"; [ $lib = prio1 ] && L="$PWD/scratch/exp_r04/libgpd_prio1.so"
  GPD_LIB=$L python bench.py --workload swarm1m_ext_240hz --steps 64 --warmup 16 --no-cpu-baseline --no-parity --min-time 0.1 2>/dev/null | tail -1 | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('$lib swarm1m us/sub-step %.3f' % (j['ms_per_step']*1e3))"
done
