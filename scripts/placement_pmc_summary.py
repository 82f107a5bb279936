# joins the launch order printed by scripts/placement_pmc.py with rocprofv3's counter_collection.csv: per pair kind, mean counter values of
# the sw_systolic2 launches.  usage: python3 scripts/placement_pmc_summary.py <log of placement_pmc.py> <dir of the rocprofv3 run>
import collections, csv, glob, json, sys
log, d = sys.argv[1], sys.argv[2]
order = None
for ln in open(log):
    if ln.startswith("PLACEMENT_PMC_ORDER "):
        order = json.loads(ln[len("PLACEMENT_PMC_ORDER "):])
assert order, "no order line in the log"
rows = []
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    rows += [r for r in csv.DictReader(open(f)) if "sw_systolic2" in r["Kernel_Name"]]
by_dispatch = collections.defaultdict(dict)
for r in rows:
    by_dispatch[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
disp = [by_dispatch[k] for k in sorted(by_dispatch)]
pos = order["warmup_fills"]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for blk in order["timed"]:
    for dd in disp[pos:pos + blk["fills"]]:
        for c, v in dd.items():
            acc[blk["pair"]][c].append(v)
    acc[blk["pair"]]["ms_per_fill (HIP events, under the profiler)"].append(blk["ms_per_fill"])
    pos += blk["fills"]
out = {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in acc.items()}
print(json.dumps(out, indent=1))
