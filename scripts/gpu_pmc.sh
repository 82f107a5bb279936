#!/bin/bash
# HBM traffic of the fill kernel from PMC counters (MI355X_MICROARCH.md, HBM section): WRITE_SIZE and FETCH_SIZE in
# SEPARATE passes (TCC slots), FETCH_SIZE doubled (gfx950 tallies 128-B read requests at 64 B), plus a calibration kernel
# of known byte count with the fill's own store pattern (one 256-B row segment per 65540-B row).  Configs 2 and 3.
export TMPDIR=/tmp
mkdir -p gpurun_out/pmc
run() {  # run <tag> <bench args...>
  local tag=$1; shift
  for c in WRITE_SIZE FETCH_SIZE; do
    timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc/${tag}_$c -- python bench.py --steps 5 --warmup 1 --no-cpu --placement-trials 1 "$@" > gpurun_out/pmc/${tag}_$c.log 2>&1
    echo "$tag $c rc=$?"
  done
}
run cfg2_16384_i32
run cfg3_65536_h64 --cols 65536 --rows 65536 --h64 --steps 2
for c in WRITE_SIZE FETCH_SIZE; do
  timeout -k 10 100 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc/calib_$c -- ./tools/ubench_store > gpurun_out/pmc/calib_$c.log 2>&1
  echo "calib $c rc=$?"
done
python3 - <<'PY' | tee gpurun_out/pmc_summary.log
import csv, glob, collections, json
vals = {}
kern = {}
for tag in ["cfg2_16384_i32", "cfg3_65536_h64", "calib"]:
    for c in ["WRITE_SIZE", "FETCH_SIZE"]:
        for f in glob.glob(f"gpurun_out/pmc/{tag}_{c}/**/*counter_collection.csv", recursive=True):
            acc = collections.defaultdict(list)
            for row in csv.DictReader(open(f)):
                acc[(row["Kernel_Name"][:60], row["Counter_Name"])].append(float(row["Counter_Value"]))
            for (k, cn), v in acc.items():
                print(f"{tag:16s} {k:60s} {cn:11s} n={len(v):3d} mean={sum(v)/len(v):14.1f} first={v[0]:14.1f} last={v[-1]:14.1f}")
                if "sw_systolic" in k:   # a fill enqueues sw_systolic2 and sw_systolic; the one that is not responsible leaves at once (~0)
                    x = sum(v[-3:]) / len(v[-3:]) if tag.startswith("cfg2") else v[-1]
                    if x > vals.get((tag, cn), 0.0):
                        vals[(tag, cn)] = x
                        kern[tag] = "sw_systolic2" if "sw_systolic2" in k else "sw_systolic"
out = {"source": "scripts/gpu_pmc.sh: rocprofv3 --pmc WRITE_SIZE / --pmc FETCH_SIZE in separate passes; counters in KiB; FETCH_SIZE doubled (gfx950)"}
for tag, key, alg in (("cfg2_16384_i32", "16384x16384 int32 H + int32 P", 16385 * 16385 * 8), ("cfg3_65536_h64", "65536x65536 int64 H + int32 P", 65537 * 65537 * 12)):
    if (tag, "WRITE_SIZE") in vals and (tag, "FETCH_SIZE") in vals:
        w, f = vals[(tag, "WRITE_SIZE")] * 1024, vals[(tag, "FETCH_SIZE")] * 1024
        out[key] = {"WRITE_SIZE_bytes": w, "FETCH_SIZE_bytes_raw": f, "traffic_bytes_per_launch": w + 2 * f, "algorithmic_bytes_per_launch": alg,
                    "traffic_over_algorithmic": (w + 2 * f) / alg, "kernel": kern.get(tag)}
print(json.dumps(out, indent=1))
open("gpurun_out/r02_pmc_traffic.json", "w").write(json.dumps(out, indent=1))
PY
