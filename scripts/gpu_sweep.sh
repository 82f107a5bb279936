#!/bin/bash
# usage: gpu_sweep.sh "<bench args 1>" "<bench args 2>" ...   -> one compact line per config
mkdir -p gpurun_out; : > gpurun_out/sweep.log
for cfg in "$@"; do
  out=$(timeout -k 10 200 python bench.py --no-cpu $cfg 2>&1 | grep '^{' | python -c '
import json,sys
for l in sys.stdin:
    d=json.loads(l); r=d["roofline"]
    print("GCUPS=%.1f ms=%.4f min_ms=%.4f frac=%.3f grid=%s strips=%s score=%s" % (d["value"], r["avg_launch_ms"], r["min_launch_ms"], r["frac"], d["config"]["grid"], d["config"]["strips"], d["config"]["max_score"]))')
  rc=$?
  echo "[$cfg] $out" | tee -a gpurun_out/sweep.log
done
