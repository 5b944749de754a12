#!/bin/bash
# round 6, validation of the tree as committed: the GPU suite (with durations), smoke(), the driver's command
O=gpurun_out/r06f; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q --durations=25 > $O/full_gpu.log 2>&1; grep -n "passed\|failed" $O/full_gpu.log | tail -2
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; tail -c 300 $O/bench_driver.json
