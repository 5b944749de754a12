#!/bin/bash
# round 6, fourth GPU call: the whole GPU suite on the restructured tree, the measured figures of the new parity tests, the drop-in latency,
# the packed-record A/B of the headline kernel, the driver's command with the placement search (and without), config 3 (ii)'s acceptance rule
O=gpurun_out/r06d; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -x -q > $O/full_gpu.log 2>&1; tail -3 $O/full_gpu.log
timeout 900 python -m pytest tests -m gpu -q -s -k "envelope or population or stack8_downwash or fixture_multihover or auto_reset_matches" > $O/measured.log 2>&1
grep -h "MEASURED\|POPULATION\|ENVELOPE\|WAKE RULE\|passed\|failed" $O/measured.log > $O/measured_summary.log
python scratch/exp_r06/dropin_latency.py > $O/dropin.log 2>&1
python scratch/exp_r06/ab_packed.py > $O/ab_packed.log 2>&1
python bench.py --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err
GPD_BENCH_NO_PLACEMENT_SEARCH=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-second-leg --no-dropin-leg --no-parity > $O/bench_noplace.json 2> $O/bench_noplace.err
python bench.py --workload stack8x8192_ext_240hz --steps 256 --warmup 64 --no-cpu-baseline --no-second-leg > $O/bench_stack8.json 2> $O/bench_stack8.err
python bench.py --workload multihover2x16384_240hz --steps 256 --warmup 64 --no-cpu-baseline --no-second-leg > $O/bench_multihover2.json 2> $O/bench_multihover2.err
du -sh $O
