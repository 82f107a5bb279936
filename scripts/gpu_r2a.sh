#!/bin/bash
# round 2, first GPU pass: new parity tests, then diagnostics for the kernel work
set -u
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest.log 2>&1; rc=$?
tail -n 25 gpurun_out/pytest.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "pytest timed out: stopping"; exit 1; fi
echo "=== strip times ns=1 nc=8 (fast producer)"
timeout -k 10 120 python scripts/strip_times.py 16384 16384 0 1 8 > gpurun_out/strip_fast.log 2>&1; tail -n 12 gpurun_out/strip_fast.log
echo "=== strip times ns=1 nc=8 generic producer (debug 4)"
timeout -k 10 120 python scripts/strip_times.py 16384 16384 4 1 8 > gpurun_out/strip_generic.log 2>&1; tail -n 12 gpurun_out/strip_generic.log
echo "=== bench first allocation"
timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu --placement-trials 1 > gpurun_out/bench_first.log 2>&1; tail -n 2 gpurun_out/bench_first.log
echo "=== bench generic producer"
timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu --placement-trials 1 --debug-flags 4 > gpurun_out/bench_generic.log 2>&1; tail -n 2 gpurun_out/bench_generic.log
echo "=== bench producer alone (debug 2)"
timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu --placement-trials 1 --debug-flags 2 > gpurun_out/bench_prodonly.log 2>&1; tail -n 2 gpurun_out/bench_prodonly.log
echo "=== bench no stores (debug 1)"
timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu --placement-trials 1 --debug-flags 1 > gpurun_out/bench_nostore.log 2>&1; tail -n 2 gpurun_out/bench_nostore.log
