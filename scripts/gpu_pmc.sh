#!/bin/bash
# HBM traffic of the fill kernel from PMC counters: separate passes per counter (TCC slots), plus a
# calibration kernel of known byte count with the same access pattern (dword per lane, 256 B per row).
export TMPDIR=/tmp
mkdir -p gpurun_out/pmc
for c in WRITE_SIZE FETCH_SIZE; do
  timeout -k 10 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc/bench_$c -- python bench.py --steps 10 --warmup 1 --no-cpu --placement-trials 1 > gpurun_out/pmc/bench_$c.log 2>&1
  echo "bench $c rc=$?"
  timeout -k 10 100 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc/calib_$c -- ./tools/ubench_store > gpurun_out/pmc/calib_$c.log 2>&1
  echo "calib $c rc=$?"
done
python3 - <<'PY'
import csv, glob, collections
for tag in ["bench_WRITE_SIZE", "bench_FETCH_SIZE", "calib_WRITE_SIZE", "calib_FETCH_SIZE"]:
    for f in glob.glob(f"gpurun_out/pmc/{tag}/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(list)
        for row in csv.DictReader(open(f)):
            acc[(row["Kernel_Name"][:60], row["Counter_Name"])].append(float(row["Counter_Value"]))
        for (k, c), v in acc.items():
            print(f"{tag:18s} {k:60s} {c:11s} n={len(v):3d} mean={sum(v)/len(v):14.1f} first={v[0]:14.1f}")
PY
