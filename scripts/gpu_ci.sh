#!/bin/bash
# Runs on the GPU box (via gpurun): parity tests, smoke, the bench lines of every BASELINE config that fits one GPU, rocprofv3
# kernel traces and PMC passes of the default bench and of the batch kernel.  A step that times out / is killed stops the chain.
# Everything lands in gpurun_out/; scripts/collect_profiles.py copies what is judged into profiles/r04_*.
set -u
mkdir -p gpurun_out
step() {  # step <name> <timeout_s> <cmd...>
    local name=$1 to=$2; shift 2
    echo "=== $name" | tee -a "$CILOG"
    timeout -k 10 "$to" "$@" > "gpurun_out/$name.log" 2>&1
    local rc=$?
    echo "$name rc=$rc" | tee -a "$CILOG"
    tail -n 2 "gpurun_out/$name.log" | cut -c1-400
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT in $name: stopping" | tee -a "$CILOG"; exit 1; fi
    return 0
}
MODE="${1:-all}"    # all | tests | bench | prof  (a gpurun call is at most 20 minutes: tests, bench and prof are one call each)
want() { [ "$MODE" = all ] || [ "$MODE" = "$1" ]; }
CILOG="gpurun_out/ci_$MODE.log"      # (one log per stage: gpurun merges a call's files over the earlier ones of the same name)
echo "--- stage $MODE, $(date -u +%FT%TZ)" > "$CILOG"
rocminfo 2>/dev/null | grep -E "Marketing Name|Compute Unit|Max Clock" | head -6 >> "$CILOG"
nproc >> "$CILOG"; lscpu | grep "Model name" >> "$CILOG"
export TMPDIR=/tmp
if want tests; then
step smoke 300 python __graft_entry__.py smoke
step pytest_gpu 1100 python -m pytest tests -m gpu -x -q --durations=8
fi
if want bench; then
# ---- BASELINE config 2 (the headline), with and without scouts / two columns per lane
step bench 300 python bench.py
step bench_no_pacing 200 python bench.py --steps 20 --warmup 3 --no-cpu --debug-flags 134217728
step bench_no_xcd_roles 200 python bench.py --steps 20 --warmup 3 --no-cpu --debug-flags 142606336
step bench_no_scouts 200 python bench.py --steps 20 --warmup 3 --no-cpu --debug-flags 131072
step bench_one_column 200 python bench.py --steps 20 --warmup 3 --no-cpu --debug-flags 16384
step bench_first_alloc 200 python bench.py --steps 20 --warmup 3 --no-cpu --placement-trials 1
step bench_trial_fill_search 200 python bench.py --steps 20 --warmup 3 --no-cpu --placement-trials 16
step bench_p8 200 python bench.py --steps 20 --warmup 3 --no-cpu --p8
# ---- other sizes
step bench_i32_8k 200 python bench.py --steps 20 --warmup 3 --no-cpu --cols 8192 --rows 8192
step bench_i32_12k 200 python bench.py --steps 20 --warmup 3 --no-cpu --cols 12288 --rows 12288
step bench_i32_18k 200 python bench.py --steps 10 --warmup 3 --no-cpu --cols 18432 --rows 18432
step bench_i32_20k 200 python bench.py --steps 10 --warmup 3 --no-cpu --cols 20480 --rows 20480
step bench_i32_24k 300 python bench.py --steps 5 --warmup 2 --no-cpu --cols 24576 --rows 24576
step bench_i32_32k 300 python bench.py --steps 5 --warmup 2 --no-cpu --cols 32768 --rows 32768
step bench_i32_32k_untiled 300 python bench.py --steps 5 --warmup 2 --no-cpu --cols 32768 --rows 32768 --debug-flags 524288
step bench_i32_40k 300 python bench.py --steps 5 --warmup 2 --no-cpu --cols 40000 --rows 40000
step bench_i32_48k 300 python bench.py --steps 3 --warmup 1 --no-cpu --cols 49152 --rows 49152
step bench_i32_64k 400 python bench.py --steps 3 --warmup 1 --no-cpu --cols 65536 --rows 65536
step bench_i32_64k_untiled 400 python bench.py --steps 3 --warmup 1 --no-cpu --cols 65536 --rows 65536 --debug-flags 524288
# ---- config 3
step bench_h64_64k 400 python bench.py --steps 3 --warmup 1 --no-cpu --cols 65536 --rows 65536 --h64
step bench_p8_64k 400 python bench.py --steps 3 --warmup 1 --no-cpu --cols 65536 --rows 65536 --p8
step bench_h64_64k_strips_every_126 400 python bench.py --steps 3 --warmup 1 --no-cpu --cols 65536 --rows 65536 --h64 --s2w 126
step bench_i32_16k_strips_every_110 200 python bench.py --steps 20 --warmup 3 --no-cpu --s2w 110
# ---- config 4 on one GPU: the per-rank shape of the 8-GPU run, the whole matrix P-only, two ranks rehearsing over gloo
step bench_band_n8_shape_p8 500 python bench.py --mode bands --cols 262144 --rows 32768 --steps 3 --warmup 1 --p8
step bench_config4_one_gpu_p_only 500 python bench.py --mode bands --cols 262144 --rows 262144 --p8 --no-h --steps 2 --warmup 1
step bench_gpus2_gloo_self_spawned 500 python bench.py --gpus 2 --backend gloo --steps 3 --warmup 1 --cols 32768 --rows 32768
step bench_gpus4_gloo_one_gpu 500 python bench.py --gpus 4 --backend gloo --steps 2 --warmup 1 --cols 32768 --rows 32768
# ---- config 5
step bench_batch_100k_scoreonly 400 python bench.py --mode batch --pairs 100000 --steps 3 --warmup 1
step bench_batch_100k_scoreonly_32bit 400 python bench.py --mode batch --pairs 100000 --steps 3 --warmup 1 --no-cpu --debug-flags 262144
step bench_batch_100k_p8 400 python bench.py --mode batch --pairs 100000 --steps 3 --warmup 1 --store --p8 --no-h --no-cpu
step bench_batch_100k_p8_32bit 400 python bench.py --mode batch --pairs 100000 --steps 3 --warmup 1 --store --p8 --no-h --no-cpu --debug-flags 2097152
step bench_batch_100k_p8_no_stores 400 python bench.py --mode batch --pairs 100000 --steps 3 --warmup 1 --store --p8 --no-h --no-cpu --debug-flags 1
step bench_batch_100k_p8_traceback 400 python bench.py --mode batch --pairs 100000 --steps 3 --warmup 1 --store --p8 --no-h --traceback --no-cpu
step bench_batch_100k_old_path 400 python bench.py --mode batch --pairs 100000 --steps 1 --warmup 1 --no-cpu --debug-flags 65536
# ---- CLI: fill + traceback timing as the reference prints them
step ubench_scope 120 ./tools/ubench_scope
step cli_16384 200 ./smith-waterman_amd/smithW 16384 16384
step cli_2bands_1gpu_16384 200 ./smith-waterman_amd/smithW --devices 0,0 16384 16384
fi
if want prof; then
# ---- rocprofv3: kernel trace + stats of the default bench and of the batch kernel
rm -rf gpurun_out/prof gpurun_out/prof_batch gpurun_out/pmc
step rocprof 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -- python bench.py --steps 20 --warmup 3 --no-cpu
step rocprof_batch 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_batch -- python bench.py --mode batch --pairs 100000 --steps 3 --warmup 1 --no-cpu
# ---- PMC passes (own runs, kernel trace only beside them)
pmc() {  # pmc <tag> <counters> <bench args...>
    local tag=$1 ctr=$2; shift 2
    step "pmc_$tag" 400 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d "gpurun_out/pmc/$tag" -- python bench.py --no-cpu "$@"
}
pmc cfg2_WRITE_SIZE WRITE_SIZE --steps 5 --warmup 1 --placement-trials 1
pmc cfg2_FETCH_SIZE FETCH_SIZE --steps 5 --warmup 1 --placement-trials 1
pmc cfg3_WRITE_SIZE WRITE_SIZE --steps 2 --warmup 1 --placement-trials 1 --cols 65536 --rows 65536 --h64
pmc cfg3_FETCH_SIZE FETCH_SIZE --steps 2 --warmup 1 --placement-trials 1 --cols 65536 --rows 65536 --h64
pmc batch_score_SQ "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES" --mode batch --pairs 20000 --steps 1 --warmup 1
pmc batch_score32_SQ "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES" --mode batch --pairs 20000 --steps 1 --warmup 1 --debug-flags 262144
pmc batch_p8_SQ "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES" --mode batch --pairs 20000 --steps 1 --warmup 1 --store --p8 --no-h
pmc batch_p8_32_SQ "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES" --mode batch --pairs 20000 --steps 1 --warmup 1 --store --p8 --no-h --debug-flags 2097152
pmc batch_p8_WRITE_SIZE WRITE_SIZE --mode batch --pairs 20000 --steps 1 --warmup 1 --store --p8 --no-h
pmc batch_p8_FETCH_SIZE FETCH_SIZE --mode batch --pairs 20000 --steps 1 --warmup 1 --store --p8 --no-h
# ---- placement: TCC write counters of the same fill into a same-class and a different-class pair
step placement_pmc 400 bash scripts/gpu_placement_pmc.sh
fi
python3 scripts/collect_profiles.py gpurun_out | tee gpurun_out/profile_summary.log
