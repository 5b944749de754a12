#!/bin/bash
O=gpurun_out/r06h; mkdir -p $O
patch -p1 < scratch/exp_r06/packed_records.patch > $O/patch.log 2>&1
python -c "from gym_pybullet_drones_amd import _native; print(_native.build(force=True))" > $O/build.log 2>&1; tail -1 $O/build.log
python scratch/exp_r06/ab_packed_hbm.py 5 > $O/ab_packed_hbm.log 2>&1; tail -6 $O/ab_packed_hbm.log
