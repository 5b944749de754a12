#!/bin/bash
# waves of the replay launch with many batches raise their priority (s_setprio): timeline + A/B + tests
GPD_LIB=$PWD/scratch/exp_r04/libgpd_tsf.so timeout 200 python scratch/exp_r04/force_timeline.py 2>&1 | grep "^slot [0-1]"
for lib in before after before after; do
  L="$PWD/gym-pybullet-drones_amd/csrc/libgpd.so"; [ $lib = before ] && L="$PWD/scratch/exp_r04/libgpd_before.so"
  GPD_LIB=$L python bench.py --workload swarm65536_ext_240hz --steps 240 --warmup 24 --no-cpu-baseline --no-parity 2>/dev/null | tail -1 | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('$lib swarm65536 us/sub-step %.3f' % (j['ms_per_step']*1e3))"
done
for lib in before after; do
  L="$PWD/gym-pybullet-drones_amd/csrc/libgpd.so"; [ $lib = before ] && L="$PWD/scratch/exp_r04/libgpd_before.so"
  GPD_LIB=$L python bench.py --workload swarm1m_ext_240hz --steps 64 --warmup 16 --no-cpu-baseline --no-parity 2>/dev/null | tail -1 | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('$lib swarm1m us/sub-step %.3f' % (j['ms_per_step']*1e3))"
done
python -m pytest tests/test_gpu_surface.py -q -k "swarm or world or stale or hipgraph or wake or halo" 2>&1 | tail -1
