#!/bin/bash
# Runs on the GPU box (via gpurun): parity tests, smoke, a short bench, and a rocprofv3 kernel trace.
# A step that times out / is killed stops the chain (no further GPU work after a hang).
set -u
mkdir -p gpurun_out
step() {  # step <name> <timeout_s> <cmd...>
    local name=$1 to=$2; shift 2
    echo "=== $name" | tee -a gpurun_out/ci.log
    timeout -k 10 "$to" "$@" > "gpurun_out/$name.log" 2>&1
    local rc=$?
    echo "$name rc=$rc" | tee -a gpurun_out/ci.log
    tail -n 15 "gpurun_out/$name.log"
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT in $name: stopping" | tee -a gpurun_out/ci.log; exit 1; fi
    return 0
}
: > gpurun_out/ci.log
rocminfo 2>/dev/null | grep -E "Marketing Name|Compute Unit|Max Clock" | head -6 >> gpurun_out/ci.log
nproc >> gpurun_out/ci.log; lscpu | grep "Model name" >> gpurun_out/ci.log
step smoke 300 python __graft_entry__.py smoke
step pytest_gpu 600 python -m pytest tests -m gpu -x -q
step bench 400 python bench.py --steps 10 --warmup 2
step bench_strip_scan 200 python bench.py --steps 5 --warmup 1 --no-cpu --engine 1
step bench_h64_32k 300 python bench.py --steps 3 --warmup 1 --no-cpu --cols 32768 --rows 32768 --h64 --placement-trials 2
step bench_h64_64k 300 python bench.py --steps 3 --warmup 1 --no-cpu --cols 65536 --rows 65536 --h64 --placement-trials 3
step bench_i32_64k 300 python bench.py --steps 3 --warmup 1 --no-cpu --cols 65536 --rows 65536 --placement-trials 3
step bench_p8 300 python bench.py --steps 10 --warmup 2 --no-cpu --p8
step bench_p8_64k 300 python bench.py --steps 3 --warmup 1 --no-cpu --cols 65536 --rows 65536 --p8 --placement-trials 3
step bench_bands 300 python bench.py --steps 3 --warmup 1 --mode bands
step bench_batch_1k 300 python bench.py --steps 3 --warmup 1 --mode batch --cols 1024 --rows 1024 --pairs 100000
step bench_batch_1k_stored 300 python bench.py --steps 3 --warmup 1 --mode batch --cols 1024 --rows 1024 --pairs 512 --store
export TMPDIR=/tmp
# the default bench command under the profiler.  The trace also holds the placement-trial and warm-up launches, so
# besides rocprofv3's own stats the average of the LAST 20 sw_systolic launches (= the timed steps) is derived from
# the kernel trace; that is the number to compare with roofline.avg_launch_ms of the JSON line this run prints.
rm -rf gpurun_out/prof
step rocprof 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -- python bench.py --steps 20 --warmup 3 --no-cpu
python3 - <<'PY' | tee gpurun_out/rocprof_timed_launches.txt
import csv, glob
for f in glob.glob("gpurun_out/prof/**/*kernel_trace.csv", recursive=True):
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in csv.DictReader(open(f)) if "sw_systolic" in r["Kernel_Name"]]
    if len(d) >= 20:
        t = d[-20:]
        print(f"{f}: {len(d)} sw_systolic launches in the trace; the last 20 (the timed steps): avg {sum(t)/20/1e6:.4f} ms, min {min(t)/1e6:.4f}, max {max(t)/1e6:.4f}; "
              f"all launches: avg {sum(d)/len(d)/1e6:.4f} ms")
PY
find gpurun_out/prof -name "*stats*" | head; for f in $(find gpurun_out/prof -name "*kernel_stats.csv"); do head -8 "$f"; done
