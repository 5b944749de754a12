#!/bin/bash
# round 6, fifth GPU call: A/B of the z-pair variant, the whole GPU suite, the driver's command plain and under rocprofv3 --kernel-trace --stats
O=gpurun_out/r06e; mkdir -p $O
python scratch/exp_r06/ab_libs.py 3 > $O/ab_pair_z.log 2>&1
timeout 1800 python -m pytest tests -m gpu -x -q > $O/full_gpu.log 2>&1; tail -3 $O/full_gpu.log
timeout 600 python -m pytest tests -m gpu -q -s -k "population" > $O/population.log 2>&1; grep -h "POPULATION\|passed\|failed" $O/population.log > $O/population_summary.log
python bench.py --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err
R=$PWD; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/$O/trace_driver --output-format csv -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $R/$O/bench_traced.json 2> $R/$O/bench_traced.err
cd $R; find $O/trace_driver -name "*kernel_trace.csv" -size +30M -delete; du -sh $O
