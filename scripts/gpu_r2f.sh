#!/bin/bash
set -u
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_multi_capi_gpu.py tests/test_band_gpu.py -m gpu -x -q 2>&1 | tail -5
bash scripts/gpu_pmc.sh 2>&1 | tail -40
