#!/bin/bash
O=gpurun_out/r06i; mkdir -p $O
for t in 0.768 0.768 0.755; do
( time python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-second-leg --no-dropin-leg --no-parity --placement-target $t > $O/bench_t$t.$RANDOM.json 2>> $O/err.log ) 2>> $O/times.txt
done
tail -12 $O/times.txt
