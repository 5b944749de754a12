#!/bin/bash
# validation of HEAD: the GPU suite, smoke(), the driver's command
O=gpurun_out/r06r; mkdir -p $O
( time python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1 ) 2>> $O/times.txt
tail -3 $O/pytest_gpu.log
( time python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" > $O/smoke.log 2>&1 ) 2>> $O/times.txt
tail -2 $O/smoke.log
( time python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err ) 2>> $O/times.txt
python - <<'P'
import json
p=json.loads(open('gpurun_out/r06r/bench_driver.json').read().strip().splitlines()[-1])
print(p['value'], p['roofline']['frac'], p['hbm_saturating']['frac'], p['hbm_saturating']['frac_range_over_allocations'], p['parity']['ok'], p['dropin_single_env']['us_per_step'])
P
cat $O/times.txt
