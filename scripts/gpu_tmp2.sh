#!/bin/bash
timeout -k 10 600 python -m pytest tests/test_xcd_roles_gpu.py -m gpu -x -q > gpurun_out/xcd_pytest.log 2>&1; rc=$?; tail -3 gpurun_out/xcd_pytest.log; [ $rc -eq 0 ] || exit 1
bash scripts/gpu_tmp.sh
