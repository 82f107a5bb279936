#!/bin/bash
set -u
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest.log 2>&1; rc=$?
tail -n 8 gpurun_out/pytest.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "pytest timed out: stopping"; exit 1; fi
if [ $rc -ne 0 ]; then echo "pytest failed rc=$rc"; grep -E "Error|error|assert" gpurun_out/pytest.log | head -20; exit 1; fi
echo "=== bench default"
timeout -k 10 400 python bench.py > gpurun_out/bench.log 2>&1; tail -n 1 gpurun_out/bench.log | cut -c1-1800
echo "=== bench bands 16384 on one GPU"
timeout -k 10 300 python bench.py --mode bands --cols 16384 --rows 16384 --steps 10 --warmup 2 > gpurun_out/bench_bands.log 2>&1; tail -n 1 gpurun_out/bench_bands.log | cut -c1-700
echo "=== CLI"
timeout -k 10 120 ./smith-waterman_amd/smithW 16384 16384 2>&1 | grep -v amdgpu | tail -4
