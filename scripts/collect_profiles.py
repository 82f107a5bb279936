#!/usr/bin/env python3
"""Summarises what scripts/gpu_ci.sh left in gpurun_out/ (bench lines, rocprofv3 kernel stats / traces, PMC passes) and, with
--copy, puts the files that are judged under profiles/r04_*.

  python3 scripts/collect_profiles.py gpurun_out            # on the GPU box: summary on stdout + gpurun_out/r04_pmc_traffic.json
  python3 scripts/collect_profiles.py gpurun_out --copy     # in the build container: copy into profiles/
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

src = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out"
copy = "--copy" in sys.argv
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
prof = os.path.join(root, "profiles")


def newest(pattern):
    """gpurun merges every call's files into the same directories: of several runs' traces only the newest counts"""
    fs = glob.glob(pattern, recursive=True)
    return [max(fs, key=os.path.getmtime)] if fs else []


def bench_line(path):
    try:
        for ln in open(path):
            if ln.startswith("{"):
                return json.loads(ln)
    except OSError:
        pass
    return None


def counters(tag, kernel_substr):
    out = collections.defaultdict(list)
    for f in newest(os.path.join(src, "pmc", tag, "**", "*counter_collection.csv")):
        for row in csv.DictReader(open(f)):
            if kernel_substr in row["Kernel_Name"]:
                out[row["Counter_Name"]].append(float(row["Counter_Value"]))
    return out


print("== bench lines")
_cilogs = [f for f in (os.path.join(src, f"ci_{m}.log") for m in ("all", "tests", "bench", "prof")) if os.path.exists(f)]
_steps = [ln[4:].strip() for f in _cilogs for ln in open(f) if ln.startswith("=== ")] if _cilogs else None
for f in sorted(glob.glob(os.path.join(src, "bench*.log"))):
    d = bench_line(f)
    if d and (_steps is None or os.path.basename(f)[:-4] in _steps):
        r = d.get("roofline", {})
        print(f"{os.path.basename(f)[:-4]:36s} {d['value'] or 0:9.1f} {d['unit']}  {d['ms_per_step'] or 0:9.4f} ms/step  n_gpus {d['n_gpus']}  "
              f"frac {r.get('frac') or 0:.3f} ({r.get('bound', '-')})  tau {r.get('tau_step_ns', 0) or 0:.1f} ns  lag {r.get('strip_handoff_lag_ns', 0) or 0:.0f} ns")

print("== rocprofv3 kernel stats (top kernels)")
for d in ("prof", "prof_batch"):
    for f in newest(os.path.join(src, d, "**", "*kernel_stats.csv")):
        for i, row in enumerate(csv.DictReader(open(f))):
            if i < 5:
                print(f"{d:11s} {row['Name'][:70]:70s} calls {row['Calls']:>5s} avg {float(row['AverageNs']) / 1e3:10.1f} us  {row['Percentage']}%")

# the timed launches of the default bench: the last launch is the stamped one (chain_stamps), the `steps` before it are timed
timed = None
for f in newest(os.path.join(src, "prof", "**", "*kernel_trace.csv")):
    rows = list(csv.DictReader(open(f)))
    d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows if "sw_systolic2" in r["Kernel_Name"]]
    if len(d) >= 21:
        t = d[-21:-1]
        timed = {"kernel": "sw_systolic2", "launches_in_trace": len(d), "timed_steps": 20, "avg_ms": sum(t) / 20 / 1e6, "min_ms": min(t) / 1e6, "max_ms": max(t) / 1e6,
                 "all_launches_avg_ms": sum(d) / len(d) / 1e6}
        # everything a fill enqueues between two big launches: the helper kernels and the gaps
        big = [r for r in rows if "sw_systolic2" in r["Kernel_Name"]][-21:-1]
        per = [(int(big[i + 1]["Start_Timestamp"]) - int(big[i]["Start_Timestamp"])) for i in range(len(big) - 1)]
        timed["start_to_start_avg_ms"] = sum(per) / len(per) / 1e6
        print("== timed launches of the default bench:", json.dumps(timed))
for f in newest(os.path.join(src, "prof_batch", "**", "*kernel_trace.csv")):
    rows = list(csv.DictReader(open(f)))
    d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows if "sw_batch_wave" in r["Kernel_Name"]]
    if d:
        print(f"== batch kernel launches: {len(d)}, avg {sum(d) / len(d) / 1e6:.3f} ms (100000 pairs of 1024x1024: {100000 * 1024 * 1024 / (sum(d) / len(d)):.0f} GCUPS in the kernel)")

print("== PMC (counter values per launch of the kernel named; WRITE_SIZE / FETCH_SIZE in KiB, FETCH_SIZE doubled for bytes: gfx950 tallies 128-B reads at 64 B)")
out = {"source": "scripts/gpu_ci.sh + scripts/collect_profiles.py: rocprofv3 --pmc, separate passes; WRITE_SIZE / FETCH_SIZE in KiB; FETCH_SIZE doubled (gfx950)"}
for key, wtag, ftag, kern, alg in (("16384x16384 int32 H + int32 P", "cfg2_WRITE_SIZE", "cfg2_FETCH_SIZE", "sw_systolic2", 16385 * 16385 * 8),
                                  ("65536x65536 int64 H + int32 P", "cfg3_WRITE_SIZE", "cfg3_FETCH_SIZE", "sw_systolic2", 65537 * 65537 * 12),
                                  ("batch 20000 x 1024x1024 int8 P", "batch_p8_WRITE_SIZE", "batch_p8_FETCH_SIZE", "sw_batch_wave", 20000 * 1025 * 1025)):
    w, fch = counters(wtag, kern).get("WRITE_SIZE"), counters(ftag, kern).get("FETCH_SIZE")
    if w and fch:
        wb, fb = max(w[-3:]) * 1024, max(fch[-3:]) * 1024
        out[key] = {"WRITE_SIZE_bytes": wb, "FETCH_SIZE_bytes_raw": fb, "traffic_bytes_per_launch": wb + 2 * fb, "algorithmic_bytes_per_launch": alg,
                    "traffic_over_algorithmic": (wb + 2 * fb) / alg, "kernel": kern}
        print(f"{key}: WRITE {wb / 1e9:.2f} GB, FETCH (x2) {2 * fb / 1e9:.2f} GB, traffic / algorithmic = {(wb + 2 * fb) / alg:.3f}")
# (steps: what all waves of the launch take together -- one wave per pair, or per two pairs in the packed kernel -- x 1088 steps each)
for tag, pairs, label, kern, per_wave in (("batch_score_SQ", 20000, "packed 16-bit, score + arg-max only", "sw_batch_wave16", 2),
                                          ("batch_score32_SQ", 20000, "score + arg-max only", "sw_batch_wave<", 1), ("batch_p8_SQ", 20000, "packed 16-bit, int8 P stored", "sw_batch_wave16", 2),
                                          ("batch_p8_32_SQ", 20000, "int8 P stored", "sw_batch_wave<", 1)):
    c = counters(tag, kern)
    if c:
        steps = pairs // per_wave * 1088
        info = {k: v[-1] for k, v in c.items()}
        out["batch kernel, " + label] = {**info, "wave_steps": steps, "pairs_per_wave": per_wave, "VALU_per_step": info.get("SQ_INSTS_VALU", 0) / steps,
                                         "SALU_per_step": info.get("SQ_INSTS_SALU", 0) / steps}
        print(f"batch kernel ({label}): VALU {info.get('SQ_INSTS_VALU', 0) / steps:.1f} + SALU {info.get('SQ_INSTS_SALU', 0) / steps:.1f} instructions per wave and step ({per_wave} x 1024 cells)")
if timed:
    out["timed launches of the default bench (rocprofv3 --kernel-trace)"] = timed
json.dump(out, open(os.path.join(src, "r04_pmc_traffic.json"), "w"), indent=1)

if copy:
    os.makedirs(prof, exist_ok=True)
    n = 0
    steps = _steps or []   # only what THIS run's stages produced
    open(os.path.join(prof, "r04_ci.log"), "w").write("".join(open(f).read() for f in _cilogs))
    for f in sorted(glob.glob(os.path.join(src, "bench*.log"))):
        d = bench_line(f)
        if d and os.path.basename(f)[:-4] in steps:
            json.dump(d, open(os.path.join(prof, "r04_" + os.path.basename(f)[:-4] + ".json"), "w"))
            n += 1
    for name, dst in (("profile_summary.log", "r04_profile_summary.log"), ("r04_pmc_traffic.json", "r04_pmc_traffic.json"),
                      ("cli_16384.log", "r04_cli.log"), ("cli_2bands_1gpu_16384.log", "r04_cli_2bands_1gpu.log"), ("pytest_gpu.log", "r04_pytest_gpu.log"),
                      ("ubench_scope.log", "r04_ubench_scope.log"), ("pmc/placement_summary.json", "r04_placement_tcc_counters_same_vs_different_class.json")):
        if os.path.exists(os.path.join(src, name)):
            shutil.copy(os.path.join(src, name), os.path.join(prof, dst)); n += 1
    for d, tag in (("prof", ""), ("prof_batch", "_batch")):
        for f in newest(os.path.join(src, d, "**", "*kernel_stats.csv")):
            shutil.copy(f, os.path.join(prof, f"r04_kernel_stats{tag}.csv")); n += 1
        for f in newest(os.path.join(src, d, "**", "*kernel_trace.csv")):
            # the trace is long (one line per launch): keep the last 400 lines
            lines = open(f).read().splitlines()
            open(os.path.join(prof, f"r04_kernel_trace{tag}.csv"), "w").write("\n".join(lines[:1] + lines[1:][-400:]) + "\n"); n += 1
    print(f"copied {n} files into profiles/")
