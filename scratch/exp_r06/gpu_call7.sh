#!/bin/bash
O=gpurun_out/r06g; mkdir -p $O
( time python bench.py --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err ) 2> $O/bench_time.txt; cat $O/bench_time.txt | tail -3
timeout 1500 python -m pytest tests/test_gpu_surface.py tests/test_gpu_c_host.py tests/test_examples.py -m gpu -x -q > $O/surface.log 2>&1; tail -2 $O/surface.log
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "fixture" > $O/fixtures.log 2>&1; tail -2 $O/fixtures.log
timeout 1500 python -m pytest tests/test_gpu_multirank.py -m gpu -x -q -k "allgather or gpus_2" > $O/multirank.log 2>&1; tail -2 $O/multirank.log
