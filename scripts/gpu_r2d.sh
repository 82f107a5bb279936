#!/bin/bash
set -u
mkdir -p gpurun_out
timeout -k 5 60 ./tools/ubench_prod_full; timeout -k 5 60 ./tools/ubench_prod_noexport
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest.log 2>&1; rc=$?
tail -n 15 gpurun_out/pytest.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "pytest timed out: stopping"; exit 1; fi
if [ $rc -ne 0 ]; then echo "pytest failed rc=$rc"; exit 1; fi
timeout -k 5 200 python scripts/strip0.py 16384,16384,0 16384,16384,2 2>&1 | grep -v amdgpu
echo "=== bench first allocation"
timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu --placement-trials 1 > gpurun_out/bench_first.log 2>&1; tail -n 1 gpurun_out/bench_first.log | cut -c1-300
