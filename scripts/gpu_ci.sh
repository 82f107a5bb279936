#!/bin/bash
# Runs on the GPU box (via gpurun): parity tests, smoke, the bench lines of every BASELINE config that fits one GPU, and a
# rocprofv3 kernel trace of the default bench.  A step that times out / is killed stops the chain.
set -u
mkdir -p gpurun_out
step() {  # step <name> <timeout_s> <cmd...>
    local name=$1 to=$2; shift 2
    echo "=== $name" | tee -a gpurun_out/ci.log
    timeout -k 10 "$to" "$@" > "gpurun_out/$name.log" 2>&1
    local rc=$?
    echo "$name rc=$rc" | tee -a gpurun_out/ci.log
    tail -n 3 "gpurun_out/$name.log" | cut -c1-600
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT in $name: stopping" | tee -a gpurun_out/ci.log; exit 1; fi
    return 0
}
: > gpurun_out/ci.log
rocminfo 2>/dev/null | grep -E "Marketing Name|Compute Unit|Max Clock" | head -6 >> gpurun_out/ci.log
nproc >> gpurun_out/ci.log; lscpu | grep "Model name" >> gpurun_out/ci.log
step smoke 300 python __graft_entry__.py smoke
step pytest_gpu 900 python -m pytest tests -m gpu -x -q
step bench 600 python bench.py
step bench_first_alloc 300 python bench.py --steps 10 --warmup 2 --no-cpu --placement-trials 1
step bench_one_column 300 python bench.py --steps 20 --warmup 3 --no-cpu --debug-flags 16384
step bench_old_producers 300 python bench.py --steps 10 --warmup 2 --no-cpu --debug-flags 16
step bench_strip_scan 200 python bench.py --steps 5 --warmup 1 --no-cpu --engine 1 --placement-trials 1
step bench_h64_64k 400 python bench.py --steps 3 --warmup 1 --no-cpu --cols 65536 --rows 65536 --h64
step bench_h64_64k_first_alloc 400 python bench.py --steps 3 --warmup 1 --no-cpu --cols 65536 --rows 65536 --h64 --placement-trials 1
step bench_h64_64k_one_column 400 python bench.py --steps 3 --warmup 1 --no-cpu --cols 65536 --rows 65536 --h64 --debug-flags 16384
step bench_i32_64k 400 python bench.py --steps 3 --warmup 1 --no-cpu --cols 65536 --rows 65536
step bench_i32_64k_first_alloc 400 python bench.py --steps 3 --warmup 1 --no-cpu --cols 65536 --rows 65536 --placement-trials 1
step bench_i32_64k_one_column 400 python bench.py --steps 3 --warmup 1 --no-cpu --cols 65536 --rows 65536 --debug-flags 16384
step bench_i32_32k 400 python bench.py --steps 5 --warmup 2 --no-cpu --cols 32768 --rows 32768
step bench_i32_32k_one_column 400 python bench.py --steps 5 --warmup 2 --no-cpu --cols 32768 --rows 32768 --debug-flags 16384
step bench_i32_8k 300 python bench.py --steps 10 --warmup 3 --no-cpu --cols 8192 --rows 8192
step bench_i32_8k_one_column 300 python bench.py --steps 10 --warmup 3 --no-cpu --cols 8192 --rows 8192 --debug-flags 16384
step bench_p8 300 python bench.py --steps 10 --warmup 2 --no-cpu --p8
step bench_bands_1gpu_16k 300 python bench.py --mode bands --cols 16384 --rows 16384 --steps 10 --warmup 2
step bench_bands_1gpu_128k_p8 600 python bench.py --mode bands --cols 131072 --rows 131072 --steps 2 --warmup 1 --p8
step bench_band_n8_shape_p8 600 python bench.py --mode bands --cols 262144 --rows 32768 --steps 3 --warmup 1 --p8
step bench_band_n8_shape_p8_one_column 600 python bench.py --mode bands --cols 262144 --rows 32768 --steps 3 --warmup 1 --p8 --debug-flags 16384
step bench_config4_one_gpu_p_only 600 python bench.py --mode bands --cols 262144 --rows 262144 --p8 --no-h --steps 2 --warmup 1
step bench_rehearsal_2ranks_gloo_1gpu 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 3 --warmup 1 --backend gloo --cols 32768 --rows 32768
step cli_16384 300 ./smith-waterman_amd/smithW 16384 16384
step cli_2bands_1gpu_16384 300 ./smith-waterman_amd/smithW --devices 0,0 16384 16384
step bench_batch_100k_scoreonly 600 python bench.py --mode batch --pairs 100000 --steps 1 --warmup 1
step bench_batch_100k_p8_traceback 900 python bench.py --mode batch --pairs 100000 --steps 1 --warmup 0 --store --p8 --no-h --traceback
step ubench_two_column_producer 100 ./tools/ubench_perm2
step ubench_lds_write_order 100 ./tools/ubench_ldsorder
export TMPDIR=/tmp
rm -rf gpurun_out/prof
step rocprof 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -- python bench.py --steps 20 --warmup 3 --no-cpu
python3 - <<'PY' | tee gpurun_out/rocprof_timed_launches.txt
import csv, glob
for f in glob.glob("gpurun_out/prof/**/*kernel_trace.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    # a fill enqueues sw_systolic2 (two columns per lane) and sw_systolic; the one that is not responsible leaves at once
    name = "sw_systolic2" if any("sw_systolic2" in r["Kernel_Name"] for r in rows) else "sw_systolic<"
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in rows if name in r["Kernel_Name"]]
    idle = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in rows if "sw_systolic<" in r["Kernel_Name"]] if name == "sw_systolic2" else []
    if len(d) >= 20:
        t = d[-21:-1]   # the last launch is the stamped one (chain_stamps); the 20 before it are the timed steps
        print(f"{f}: {len(d)} {name} launches in the trace; the 20 timed steps: avg {sum(t)/20/1e6:.4f} ms, min {min(t)/1e6:.4f}, max {max(t)/1e6:.4f}; "
              f"all launches: avg {sum(d)/len(d)/1e6:.4f} ms" + (f"; the {len(idle)} sw_systolic launches beside them (not responsible, leave at once): avg {sum(idle)/len(idle)/1e3:.1f} us" if idle else ""))
PY
for f in $(find gpurun_out/prof -name "*kernel_stats.csv"); do head -8 "$f"; done
