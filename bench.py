#!/usr/bin/env python3
"""bench.py -- GCUPS of the Smith-Waterman DP fill on MI355X (BASELINE.json metric).

One "step" = one full DP fill (H and P matrices + arg-max) of ONE cols x rows random DNA pair whose sequences are
already resident in HBM.

  N = 1   BASELINE config[1]: 16384 x 16384, int32 H + int32 P on one GPU.  `value` is measured on output buffers from
          the C-ABI allocator sw_alloc_outputs (what a C caller gets); `config.value_first_allocation` / `value_foreign_pair` are the same fill
          into a plain first allocation.
  N > 1   BASELINE config[3]: ONE 262144 x 262144 matrix cut into N row bands, one per rank (strong scaling: the work
          is fixed, N varies), band-resident launches with the halo rows forwarded rank to rank over RCCL.  550 GB of
          int32 H + int32 P do not fit two GPUs, so every N > 1 point uses int32 H + int8 P (5 B/cell, 344 GB in total:
          172 GB per GPU at N = 2); `config.workload` says so.  (--mode replicas keeps the old one-pair-per-rank run.)

Prints ONE JSON line on rank 0.
"""
import argparse
import importlib
import json
import os
import re
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
if int(os.environ.get("WORLD_SIZE", "1")) > 1:
    # several ranks: a band kernel stays resident while RCCL's send/recv kernels (other streams) deliver its halo.  Give every
    # stream its own hardware queue (the HIP default of 4 lets a transfer queue up behind the band kernel), and keep
    # RCCL's point-to-point kernels within the CUs the band launch leaves free (--reserve-cus).  Set before HIP initialises.
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    os.environ.setdefault("NCCL_MAX_P2P_NCHANNELS", "4")
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def _best_of(cmd, n, env=None, timeout=120):
    best = None
    for _ in range(n):
        t = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env).stdout
        m = re.search(r"scoring matrix computation:\s*([0-9.]+)", t)
        if m:
            s = float(m.group(1))
            best = s if best is None else min(best, s)
    return best


def _clocks():
    """Current sclk / mclk of the GPUs this process can see (sysfs; the entry marked * is the active level)."""
    out = {}
    import glob
    for kind in ("sclk", "mclk"):
        vals = []
        for f in sorted(glob.glob(f"/sys/class/drm/card*/device/pp_dpm_{kind}")):
            try:
                cur = [ln for ln in open(f).read().splitlines() if ln.rstrip().endswith("*")]
                if cur:
                    vals.append(cur[0].split(":")[1].replace("*", "").strip())
            except OSError:
                pass
        out[kind] = vals
    return out


def cpu_baseline(cols, rows):
    """The REAL reference programs (oracle/_ref, built from /root/reference in the build container) timed on this box's
    host cores; falls back to the C port (oracle/liboracle.so).  Every leg is bounded (a few seconds): serial_smithW on the
    full workload best of 3 like run-v1.sh:33, the OpenMP variants on bounded samples."""
    ref = os.path.join(ROOT, "oracle", "_ref", "serial_smithW")
    out = {}
    ncores = len(os.sched_getaffinity(0))
    if os.path.exists(ref):
        sec = _best_of([ref, str(cols), str(rows)], 3)
        out = {"value": cols * rows / sec / 1e9, "unit": "GCUPS", "cores": 1, "kind": "reference",
               "sample": f"serial_smithW {cols} {rows} (the full workload, fill loop only), best of 3: {sec:.3f} s"}
        omp = os.path.join(ROOT, "oracle", "_ref", "omp_smithW-v1")
        if os.path.exists(omp):
            # one barrier per anti-diagonal: more threads are not faster.  Best of 3 with 16 threads; with every hardware thread
            # (what the reference's own scripts use, run-v1.sh:33) the same program is ~1000x slower: ONE run of a 1024^2 sample
            env = dict(os.environ, OMP_NUM_THREADS="16", OMP_PROC_BIND="close")
            sec = _best_of([omp, str(cols), str(rows)], 3, env)
            if sec:
                out["omp"] = {"value": cols * rows / sec / 1e9, "unit": "GCUPS", "cores": 16, "kind": "reference",
                              "sample": f"omp_smithW-v1-refinedOrig -DSKIP_BACKTRACK {cols} {rows}, 16 threads, best of 3: {sec:.3f} s"}
            if ncores > 16:
                n = min(cols, 1024)
                env = dict(os.environ, OMP_NUM_THREADS=str(ncores), OMP_PROC_BIND="close")
                try:
                    sec = _best_of([omp, str(n), str(n)], 1, env, timeout=20)
                except subprocess.TimeoutExpired:
                    sec = None
                if sec:
                    out["omp_all_cores"] = {"value": n * n / sec / 1e9, "unit": "GCUPS", "cores": ncores, "kind": "reference",
                                            "sample": f"omp_smithW-v1-refinedOrig -DSKIP_BACKTRACK {n} {n}, all {ncores} hardware threads: {sec:.3f} s"}
        ompc = os.path.join(ROOT, "oracle", "_ref", "omp_smithW")
        if os.path.exists(ompc):
            # omp_smithW.c takes an `omp critical` per cell (omp_smithW.c:384-387): a bounded sample, 2048 x 2048
            env = dict(os.environ, OMP_NUM_THREADS=str(min(ncores, 16)), OMP_PROC_BIND="close")
            n = min(cols, 2048)
            try:
                sec = _best_of([ompc, str(n), str(n)], 1, env, timeout=20)
            except subprocess.TimeoutExpired:
                sec = None
            if sec:
                out["omp_critical"] = {"value": n * n / sec / 1e9, "unit": "GCUPS", "cores": min(ncores, 16), "kind": "reference",
                                       "sample": f"omp_smithW (per-cell critical arg-max) {n} {n}, {min(ncores, 16)} threads: {sec:.3f} s"}
    else:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_lib
        orc = oracle_lib.Oracle()
        a, b = orc.generate(cols, rows, 1)
        t0 = time.time()
        orc.fill_streaming(a, b)
        sec = time.time() - t0
        out = {"value": cols * rows / sec / 1e9, "unit": "GCUPS", "cores": 1, "kind": "port",
               "sample": f"oracle streaming fill {cols}x{rows}, {sec:.3f} s"}
    return out


def chain_stamps(sw, eng, torch, d_a, d_b, out, cols, rows):
    """One extra fill with the kernel's debug stamps: the producer's time per anti-diagonal step (strip 0 never waits
    for a neighbour) and the lag per strip hand-off, for the dependency bound N^2 / ((2N-1) tau)."""
    import numpy as np
    S = (cols + 62) // 63
    dbg = torch.zeros(16 * S + 256, dtype=torch.int64, device=out.res.device)
    eng.set_option("debug_buf", dbg.data_ptr())
    eng.fill_into(out, d_a, d_b)
    eng.synchronize()
    eng.set_option("debug_buf", 0)
    S2 = int(eng.get_option("last_strips2"))   # > 0: the two-column kernel ran (126 columns per strip)
    if S2 > 0:
        S = S2
    raw = dbg.cpu().numpy()
    nsc = int(eng.get_option("last_scouts")) if S2 > 0 else 0
    # with scouts (sw_systolic2.inc) the chain of strips is theirs: their stamps sit at [0, 2 (S-1)) -- the last strip has no scout --
    # and those of the workgroups that write the matrices ("fillers") at offset 6 S + 32
    NS = S - 1 if nsc else S
    t = raw[:2 * NS].reshape(NS, 2).astype(np.float64) * 10.0   # 100 MHz ticks -> ns
    steps = rows + 63 + (S - 1)
    tau = (t[0, 1] - t[0, 0]) / max(1, steps)
    lag = float(np.diff(t[:, 1]).mean()) if NS > 1 else 0.0
    # shader clock while the kernel ran: s_memtime ticks per s_memrealtime tick (100 MHz), stamped around strip 0's producer
    ghz = None
    if S2 > 0 and raw[6 * S + 17] > raw[6 * S + 16] and t[0, 1] > t[0, 0]:
        ghz = float(raw[6 * S + 17] - raw[6 * S + 16]) / (t[0, 1] - t[0, 0])
    extra = {"scout_workgroups": nsc}
    if nsc:
        f = raw[6 * S + 32: 6 * S + 32 + 2 * S].reshape(S, 2).astype(np.float64) * 10.0
        extra["filler_step_ns"] = float((f[0, 1] - f[0, 0]) / max(1, steps))   # (strip 0's filler never waits for a halo)
        extra["last_filler_after_last_scout_us"] = float((f[:, 1].max() - t[:, 1].max()) / 1e3)
        extra["chain_end_to_end_us"] = float((t[:, 1].max() - t[0, 0]) / 1e3)
    return tau, lag, ghz, extra


def _pmc_file():
    for name in ("r04_pmc_traffic.json", "r03_pmc_traffic.json", "r02_pmc_traffic.json"):
        f = os.path.join(ROOT, "profiles", name)
        if os.path.exists(f):
            return f
    return None


def traffic_for(workload_key, kernel=None):
    """HBM bytes per launch from the COMMITTED PMC passes (scripts/gpu_ci.sh, rocprofv3 --pmc WRITE_SIZE / FETCH_SIZE in separate
    runs) -- not measured in this run: returns (bytes, where they come from)."""
    try:
        f = _pmc_file()
        pm = json.load(open(f)).get(workload_key, {})
        if kernel and pm.get("kernel") and pm["kernel"] != kernel:
            return None, None
        return pm.get("traffic_bytes_per_launch"), f"profiles/{os.path.basename(f)} (rocprofv3 --pmc WRITE_SIZE + 2 x FETCH_SIZE of an earlier run of this workload; not measured in this run)"
    except Exception:
        return None, None


def batch_valu_per_step(label):
    """VALU instructions one wave issues per step of the batch kernel (a row of its pair -- or of its two pairs -- x 1024 columns), from the
    committed PMC pass (SQ_INSTS_VALU per launch / steps, scripts/collect_profiles.py); None when there is no such pass."""
    try:
        f = _pmc_file()
        e = json.load(open(f)).get(label)
        return (e["VALU_per_step"], f"profiles/{os.path.basename(f)}: SQ_INSTS_VALU / (waves x steps) of '{label}'") if e else (None, None)
    except Exception:
        return None, None


def run_pair(args, sw, eng, torch, dist, rank, world, local):
    cols, rows = args.cols, args.rows
    a, b = sw.generate(cols, rows, 1 + rank)          # reference generator; rank r uses seed 1+r (replicas mode)
    d_a, _ = eng.to_device(a)
    d_b, _ = eng.to_device(b)
    h_dtype = torch.int64 if args.h64 else torch.int32
    p_dtype = torch.int8 if args.p8 else None

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(out, steps, warmup):
        for _ in range(warmup):
            eng.fill_into(out, d_a, d_b)
        barrier()
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
        t0 = time.perf_counter()
        for e0, e1 in evs:
            e0.record()
            eng.fill_into(out, d_a, d_b)
            e1.record()
        barrier()
        dt = time.perf_counter() - t0
        return dt, [e0.elapsed_time(e1) for e0, e1 in evs]

    clocks0 = _clocks()
    t_alloc = 0.0
    if args.placement_trials == 1:
        # a plain first allocation: what a caller of sw_device_malloc gets
        out, placement_ms = eng.alloc(cols, rows, h_dtype, p_dtype), None
    else:
        # the C-ABI allocator: candidate placements classified by its two-stream store probe (or, --placement-trials N, tried with real
        # fills), outside the timed region and before anything else touches the memory
        eng.set_option("placement_budget_ms", 20000)   # (setup, outside the timed region: the pair is filled into many times)
        eng.set_option("placement_hold_gib", 48)        # (... and may hold slack beside P where the box's free memory is all of one class)
        t_alloc = time.perf_counter()
        out, placement_ms = eng.alloc_outputs(d_a, d_b, cols, rows, h_dtype, p_dtype, trials=args.placement_trials)
        t_alloc = time.perf_counter() - t_alloc
    # (read now: the plain pair of the `value_first_allocation` leg below is probed too)
    out_ratio = eng.get_option("last_placement_ratio_x1000") / 1000 if args.placement_trials != 1 else None
    out_held = eng.get_option("last_placement_held_gib") if args.placement_trials != 1 else 0
    # pre-heat: at least --preheat seconds of back-to-back fills before anything is timed (a freshly leased GPU idles at
    # 95 MHz; the W warm-up steps of the contract follow, inside timed()) -- into the buffers of the timed region, so that a
    # kernel trace of this command averages launches of ONE kind
    t_heat = time.perf_counter()
    nheat = 0
    while time.perf_counter() - t_heat < args.preheat:
        for _ in range(20):
            eng.fill_into(out, d_a, d_b)
        torch.cuda.synchronize()
        nheat += 20
    f0, f1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    out.H.fill_(0); torch.cuda.synchronize()
    f0.record()
    for _ in range(5):
        out.H.fill_(0)
    f1.record(); torch.cuda.synchronize()
    fill_gbs = 5 * out.H.numel() * out.H.element_size() / (f0.elapsed_time(f1) * 1e-3) / 1e9
    # the same fill into plain pairs (a handful of launches each): `value_first_allocation` -- two back-to-back allocations through the C-ABI
    # (sw_alloc_outputs with trials = 1: no search, but the library probes the pair once and picks the strip geometry by the class it
    # finds) -- and `value_foreign_pair` -- two torch allocations the library knows nothing about
    if args.placement_trials == 1:
        first = out
    else:
        first, _ = eng.alloc_outputs(d_a, d_b, cols, rows, h_dtype, p_dtype, trials=1)
    first_ratio = eng.get_option("last_placement_ratio_x1000") / 1000 if first is not out else None
    dt_first, ms_first = timed(first, max(3, args.steps // 4), 4)
    value_first = world * len(ms_first) * cols * rows / dt_first / 1e9
    first_strips = int(eng.get_option("last_strips2")) or eng.get_option("last_strips")
    if first is not out:
        first.free()
    foreign = eng.alloc(cols, rows, h_dtype, p_dtype)
    dt_foreign, ms_foreign = timed(foreign, max(3, args.steps // 4), 4)
    value_foreign = world * len(ms_foreign) * cols * rows / dt_foreign / 1e9
    del foreign
    torch.cuda.empty_cache()
    dt, kern_ms = timed(out, args.steps, args.warmup)
    res = out.result()
    tau_ns, lag_ns, shader_ghz, chain_extra = chain_stamps(sw, eng, torch, d_a, d_b, out, cols, rows) if args.engine == 0 else (0.0, 0.0, None, {})
    clocks1 = _clocks()
    if world > 1:
        tmax = torch.tensor([dt], device=f"cuda:{local}")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    if rank != 0:
        return
    cells = cols * rows
    bytes_per_cell = (8 if args.h64 else 4) + (1 if args.p8 else 4)   # mandatory H + P output only
    avg_ms = sum(kern_ms) / len(kern_ms)
    achieved = bytes_per_cell * cells / (avg_ms * 1e-3) / 1e9
    key = f"{cols}x{rows} {'int64' if args.h64 else 'int32'} H + {'int8' if args.p8 else 'int32'} P"
    traffic, traffic_src = traffic_for(key, "sw_systolic2" if int(eng.get_option("last_strips2")) > 0 else "sw_systolic") if args.engine == 0 else (None, None)
    line = {
        "metric": "GCUPS (DP cell updates/s) on NxN random DNA pair", "value": world * args.steps * cells / dt / 1e9,
        "unit": "GCUPS", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "int64" if args.h64 else "int32", "data": "synthetic",
        "config": {"workload": f"{cols}x{rows} random DNA pair (reference generator, seed 1+rank), linear gap 3/-3/-2, "
                               f"{'int64' if args.h64 else 'int32'} H + {'int8' if args.p8 else 'int32'} P written to HBM, arg-max tracked",
                   "per_gpu": "one pair per GPU" + (" (replicas)" if world > 1 else ""), "max_pos": res["max_pos"], "max_score": res["max_score"],
                   "grid": eng.get_option("last_grid"), "column_tiles": eng.get_option("last_tiles"), "strips": int(eng.get_option("last_strips2")) or eng.get_option("last_strips"),
                   "output_buffers": ("sw_alloc_outputs (C-ABI allocator: H and P in different classes of the HBM, " +
                                      ("candidates classified by a two-stream store probe, no trial fills)" if args.placement_trials <= 0 else "placement chosen by trial fills)"))
                                     if placement_ms is not None else "plain first allocation",
                   "placement_probe_ms" if args.placement_trials <= 0 else "placement_trials_ms": placement_ms, "sw_alloc_outputs_ms": t_alloc * 1e3,
                   "placement_held_gib": out_held,
                   "placement_ratio": out_ratio if placement_ms is not None else None,
                   "value_first_allocation": value_first, "value_first_allocation_is": "the same fill into a plain pair from the C-ABI (sw_alloc_outputs, trials = 1: two back-to-back hipMallocs, usually one class of the HBM; the library probes the pair once and fills a one-class pair with overlapping strips)",
                   "first_allocation_probe_ratio": first_ratio, "first_allocation_strips": first_strips,
                   "value_foreign_pair": value_foreign, "value_foreign_pair_is": "the same fill into two torch allocations (not probed: 126-column strips whatever their class)",
                   "preheat_fills": nheat,
                   "clocks_before": clocks0, "clocks_after": clocks1, "shader_clock_ghz_in_kernel": shader_ghz,
                   "ms_first_allocation": sum(ms_first) / len(ms_first)},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS,
                     "traffic": traffic, "traffic_source": traffic_src,
                     "kernel": ("sw_systolic2" if int(eng.get_option("last_strips2")) > 0 else "sw_systolic") if args.engine == 0 else "sw_strip_scan", "avg_launch_ms": avg_ms, "min_launch_ms": min(kern_ms),
                     "algorithmic_bytes_per_cell": bytes_per_cell, "measured_contiguous_fill_GBs": fill_gbs,
                     "tau_step_ns": tau_ns, "strip_handoff_lag_ns": lag_ns, **chain_extra,
                     # the chain of sequential steps with free hand-offs: rows + columns / (columns per lane) steps of tau each
                     "dependency_bound_gcups": (cells / ((rows + cols / (2 if int(eng.get_option("last_strips2")) > 0 else 1) - 1) * tau_ns)) if tau_ns > 0 else None},
    }
    if not args.no_cpu and world == 1:
        cb = cpu_baseline(cols, rows)
        for k in ("omp", "omp_all_cores", "omp_critical"):
            v = cb.pop(k, None)
            if v:
                line["cpu_baseline_" + k] = v
        line["cpu_baseline"] = cb
    print(json.dumps(line), flush=True)


def run_bands(args, sw, eng, torch, dist, rank, world, local):
    """ONE matrix as `world` row bands, band-resident launches (sw_fill_band_device), halo rows rank to rank."""
    import numpy as np
    multi = importlib.import_module("smith-waterman_amd.multi")
    cols, rows = args.cols, args.rows
    p8 = args.p8 or (world > 1 and not args.p32)
    a, b = sw.generate(cols, rows, 1)
    pipe = multi.BandResident(dist, rank, world, eng, a, b, nchunks=args.chunks, p_dtype=torch.int8 if p8 else None,
                              want_h=not args.no_h, reserve_cus=args.reserve_cus)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        pipe.fill()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        score, pos = pipe.fill()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], device=f"cuda:{local}")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    if rank != 0:
        return
    cells = cols * rows
    bpc = (0 if args.no_h else 4) + (1 if p8 else 4)
    achieved = bpc * cells / (dt / args.steps) / 1e9       # whole job; per GPU = / world
    line = {"metric": "GCUPS (DP cell updates/s) on NxN random DNA pair", "value": args.steps * cells / dt / 1e9, "unit": "GCUPS",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "int32", "data": "synthetic",
            "config": {"workload": f"ONE {cols}x{rows} random DNA pair (reference generator, seed 1), linear gap 3/-3/-2, as {world} row band(s) of "
                                   f"{-(-rows // world)} rows, one band-resident launch per GPU, "
                                   f"{'no H' if args.no_h else 'int32 H'} + {'int8' if p8 else 'int32'} P written to HBM ({bpc} B/cell, "
                                   f"{bpc * (cols + 1) * (rows + 1) / world / 2**30:.0f} GiB per GPU), arg-max tracked, halo rows as granules over "
                                   f"{('RCCL send/recv' if pipe.nccl else 'gloo send/recv (rehearsal: ranks share GPUs)') if world > 1 else 'nothing (single band)'} in {len(pipe.chunks)} column chunks",
                       "mode": "bands", "max_score": score, "max_pos": pos,
                       "placement_trials_ms_rank0": [round(x, 3) for x in getattr(pipe, "placement_ms", [])]},
            "roofline": {"bound": "hbm", "achieved": achieved / world, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / world / HBM_PEAK_GBS,
                         "traffic": None, "kernel": "sw_systolic2" if int(eng.get_option("last_strips2")) > 0 else "sw_systolic",
                         "algorithmic_bytes_per_cell": bpc, "per": "GPU (whole job / n_gpus)"}}
    print(json.dumps(line), flush=True)


VALU_PEAK_GWIPS = 1024 * 2.4 / 4.0   # 256 CUs x 4 SIMDs, one wave64 VALU instruction per 4 clk per SIMD at 2.4 GHz (G wave-instr/s)


def run_batch(args, sw, eng, torch, dist, rank, world, local):
    import numpy as np
    cols, rows, npairs = args.cols, args.rows, args.pairs
    gen = [sw.generate(cols, rows, 1 + rank * npairs + k) for k in range(npairs)]
    A, B = np.stack([g[0] for g in gen]), np.stack([g[1] for g in gen])
    p_dtype = torch.int8 if args.p8 else None
    d_a, d_b, _, _ = eng.batch_to_device(A, B)          # the sequences are resident in HBM before anything is timed
    out = None

    def step():
        nonlocal out
        r = eng.batch_device(d_a, d_b, cols, rows, store=args.store, p_dtype=p_dtype, store_h=not args.no_h, traceback=args.traceback, out=out)
        out = r[:3]
        return r

    for _ in range(max(1, args.warmup)):
        step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(args.steps):
        step()
    e1.record()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if rank != 0:
        return
    cells = npairs * cols * rows
    bpc = ((0 if args.no_h else 4) + (1 if args.p8 else 4)) if args.store else 0
    bk = eng.get_option("last_batch_kernel")
    wave, packed = bk >= 1, bk == 2
    what = f"{npairs} independent {cols}x{rows} pairs (pair k seeded 1+k) in one call, sequences resident in HBM: "
    what += (f"{'int32 H + ' if not args.no_h else ''}{'int8' if args.p8 else 'int32'} P stored ({bpc} B/cell)" if args.store else "score + exact maxPos only (no matrices)")
    what += ", per-pair traceback" if args.traceback else ""
    line = {"metric": "GCUPS (DP cell updates/s)", "value": args.steps * cells / dt / 1e9, "unit": "GCUPS", "n_gpus": 1, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "int16" if packed else "int32", "data": "synthetic",
            "config": {"workload": what, "mode": "batch", "kernel": ("sw_batch_wave16 (two pairs per wave, packed 16-bit lanes)" if packed else "sw_batch_wave (one pair per wave)") if wave else "sw_systolic (single-pair machinery)",
                       "device_ms_per_step": e0.elapsed_time(e1) / args.steps}}
    if wave:
        pb = (1 if args.p8 else 4) if args.store else 0
        if packed:
            label = "batch kernel, packed 16-bit, int8 P stored" if pb == 1 else "batch kernel, packed 16-bit, score + arg-max only"
        else:
            label = "batch kernel, int8 P stored" if pb == 1 else ("batch kernel, score + arg-max only" if pb == 0 else None)
        ipstep, isrc = batch_valu_per_step(label) if label else (None, None)
        wsteps = (npairs // 2 if packed else npairs) * (rows + 64) * -(-cols // 1024)     # steps all waves together take
        line["roofline"] = {"bound": "valu", "achieved": None, "peak": VALU_PEAK_GWIPS, "unit": "G wave-instr/s", "frac": None, "traffic": None,
                            "kernel": f"sw_batch_wave16<LE4, K12, {'true' if pb == 1 else 'false'}>" if packed else f"sw_batch_wave<16,{pb}>",
                            "note": "integer max/add recurrence: no contraction, so no MFMA; the bound is vector issue (1 wave64 instruction per 4 clk per SIMD)"}
        if ipstep:
            ach = wsteps * ipstep / (dt / args.steps) / 1e9
            line["roofline"].update({"achieved": ach, "frac": ach / VALU_PEAK_GWIPS, "valu_instructions_per_wave_step": ipstep, "valu_instructions_source": isrc})
        if bpc:
            hb = bpc * cells / (dt / args.steps) / 1e9
            line["roofline"]["hbm"] = {"achieved": hb, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": hb / HBM_PEAK_GBS, "algorithmic_bytes_per_cell": bpc}
            if packed:   # (the packed kernel with P is bound by its P stream, not by vector issue: DESIGN.md section 9)
                line["roofline"]["note"] += "; with P stored the packed kernel is bound by the store stream (see hbm)"
    elif bpc:
        ach = bpc * cells / (dt / args.steps) / 1e9
        line["roofline"] = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": None,
                            "kernel": "sw_systolic", "algorithmic_bytes_per_cell": bpc}
    if not args.no_cpu:
        ref = os.path.join(ROOT, "oracle", "_ref", "serial_smithW")
        if os.path.exists(ref):   # the reference handles one pair per process: serial_smithW <cols> <rows>, best of 5 (fill loop only)
            sec = _best_of([ref, str(cols), str(rows)], 5)
            line["cpu_baseline"] = {"value": cols * rows / sec / 1e9, "unit": "GCUPS", "cores": 1, "kind": "reference",
                                    "sample": f"serial_smithW {cols} {rows} (one pair of the batch), best of 5: {sec * 1e3:.2f} ms per pair"}
    print(json.dumps(line), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--cols", type=int, default=0)
    ap.add_argument("--rows", type=int, default=0)
    ap.add_argument("--h64", action="store_true", help="int64 H (BASELINE config 3 element type)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--mode", default="auto", choices=["auto", "pair", "bands", "batch", "replicas"],
                    help="auto: pair on one GPU, bands (one matrix sharded over the ranks) on several; replicas: one independent pair "
                         "per GPU; batch: --pairs independent pairs in one call (BASELINE config 5)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="several ranks: nccl (= RCCL; one GPU per rank) or gloo (rehearsal, ranks may share a GPU)")
    ap.add_argument("--chunks", type=int, default=64, help="bands: column chunks the halo row is forwarded in")
    ap.add_argument("--reserve-cus", type=int, default=16, help="bands, several GPUs: CUs left to the halo transfers")
    ap.add_argument("--p32", action="store_true", help="bands on several GPUs: int32 P instead of int8 (550 GB at 262144^2)")
    ap.add_argument("--no-h", action="store_true", help="bands / batch: do not write H (P-only)")
    ap.add_argument("--pairs", type=int, default=512, help="batch mode: pairs")
    ap.add_argument("--store", action="store_true", help="batch mode: also write the matrices of every pair")
    ap.add_argument("--traceback", action="store_true", help="batch mode: per-pair traceback (needs --store)")
    ap.add_argument("--engine", type=int, default=0, help="0 systolic (default), 1 strip_scan")
    ap.add_argument("--ns", type=int, default=0, help="systolic: strips per workgroup")
    ap.add_argument("--nc", type=int, default=0, help="systolic: consumer waves per strip")
    ap.add_argument("--p8", action="store_true", help="compact predecessor matrix: int8 P")
    ap.add_argument("--placement-trials", type=int, default=0,
                    help="pair mode: 0 = sw_alloc_outputs classifies candidate placements with its store probe (default, no trial fills), "
                         "1 = plain allocation, > 1 = round 3's search with that many trial fills")
    ap.add_argument("--store-policy", type=int, default=0, help="systolic H/P stores: 0 auto, 1 write-back, 2 streaming")
    ap.add_argument("--importers", type=int, default=0)
    ap.add_argument("--xcd-order", type=int, default=0, help="systolic: 1 = neighbouring strip groups on one XCD")
    ap.add_argument("--debug-flags", type=int, default=0)
    ap.add_argument("--s2w", type=int, default=0, help="two-column kernel: strips every 126 or 110 columns (0: the library chooses)")
    ap.add_argument("--max-blocks", type=int, default=0)
    ap.add_argument("--batch-lds", type=int, default=0, help="batch kernel: dynamic LDS bytes per workgroup (caps the waves per CU; experiments)")
    ap.add_argument("--preheat", type=float, default=0.5, help="pair mode: seconds of untimed back-to-back fills before the warm-up steps")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: start the N ranks ourselves, as CHILD processes, before this process has
        # touched the GPU (it never does), and leave with their exit code
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    ndev = torch.cuda.device_count()
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: refusing to report a line for another GPU count", file=sys.stderr)
        sys.exit(2)
    if world > 1 and args.backend == "nccl" and ndev < world:
        print(f"bench.py: --gpus {world} over RCCL needs {world} GPUs, this box has {ndev} (--backend gloo rehearses with ranks sharing GPUs)", file=sys.stderr)
        sys.exit(3)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local}"))
        else:   # rehearsal of the multi-rank path on a box with fewer GPUs than ranks: gloo, the ranks share GPUs
            dist.init_process_group("gloo")
            local = local % max(1, ndev)
    torch.cuda.set_device(local)
    sw = importlib.import_module("smith-waterman_amd")
    eng = sw.Engine(local)
    if world > 1 and args.backend != "nccl" and ndev < world and not args.max_blocks:
        # persistent launches of several ranks on one GPU must all be resident: split the CUs
        args.max_blocks = max(8, eng.get_option("num_cus") // -(-world // max(1, ndev)) - 16)
    eng.set_option("engine", args.engine)
    for name, v in (("importers", args.importers), ("store_policy", args.store_policy), ("strips_per_group", args.ns), ("consumers", args.nc),
                    ("debug_flags", args.debug_flags), ("max_blocks", args.max_blocks), ("xcd_order", args.xcd_order), ("batch_lds", args.batch_lds), ("s2w", args.s2w)):
        if v:
            eng.set_option(name, v)
    mode = args.mode
    if mode == "auto":
        mode = "pair" if world == 1 else "bands"
    if mode == "replicas":
        mode = "pair"
    if not args.cols:
        args.cols = 262144 if mode == "bands" and world > 1 else (1024 if mode == "batch" else 16384)
    if not args.rows:
        args.rows = args.cols
    try:
        {"pair": run_pair, "bands": run_bands, "batch": run_batch}[mode](args, sw, eng, torch, dist, rank, world, local)
    finally:
        eng.close()
        if world > 1:
            dist.destroy_process_group()


if __name__ == "__main__":
    main()
