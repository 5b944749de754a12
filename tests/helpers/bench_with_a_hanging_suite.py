#!/usr/bin/env python3
"""bench.py with a suite whose first workload never returns (every rank sleeps in its builder: a collective that hangs) -- what
tests/test_gpu_multirank.py::test_bench_watchdog_prints_the_headline_when_the_suite_hangs runs.  The hang lives HERE, not behind an
environment variable in the driver-run file."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402


def never(w, device, seed, **kw):
    time.sleep(3600)


bench.WORKLOADS["hangs_forever"] = dict(bench.WORKLOADS["hover4096_240hz"], builder=never)
bench.SUITE = ("hangs_forever",) + tuple(bench.SUITE)
if __name__ == "__main__":
    bench.__file__ = os.path.abspath(__file__)        # (the self-launched ranks run THIS file)
    bench.main(sys.argv[1:])
