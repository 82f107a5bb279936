#!/bin/bash
set -u
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest.log 2>&1; rc=$?
tail -n 25 gpurun_out/pytest.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "pytest timed out: stopping"; exit 1; fi
if [ $rc -ne 0 ]; then echo "pytest failed rc=$rc"; exit 1; fi
echo "=== bench first allocation (perm)"
timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu --placement-trials 1 > gpurun_out/bench_first.log 2>&1; tail -n 2 gpurun_out/bench_first.log
echo "=== bench first allocation (old fast, debug 16)"
timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu --placement-trials 1 --debug-flags 16 > gpurun_out/bench_old.log 2>&1; tail -n 2 gpurun_out/bench_old.log
echo "=== strip times perm"
timeout -k 10 120 python scripts/strip_times.py 16384 16384 0 1 8 > gpurun_out/strip_perm.log 2>&1; tail -n 12 gpurun_out/strip_perm.log
