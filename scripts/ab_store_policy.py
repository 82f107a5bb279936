# store policy (1 write-back, 2 streaming) x strip geometry (126 / 110) of the two-column kernel on the same buffers: usage  ab_store_policy.py <n>[h] ...
import importlib, sys, torch
sys.path.insert(0, ".")
sw = importlib.import_module("smith-waterman_amd")
eng = sw.Engine(0); eng.set_option("placement_budget_ms", 20000)
for x in sys.argv[1:] or ["65536h"]:
    n, h64 = int(x.rstrip("h")), x.endswith("h")
    a, b = sw.generate(n, n, 1); d_a, _ = eng.to_device(a); d_b, _ = eng.to_device(b)
    out, ms = eng.alloc_outputs(d_a, d_b, n, n, torch.int64 if h64 else torch.int32)
    for w in (126, 110):
        for pol in (1, 2):
            eng.set_option("store_policy", pol); eng.set_option("s2w", w)
            reps = 20 if n < 30000 else 3
            for _ in range(60 if n < 30000 else 1): eng.fill_into(out, d_a, d_b)
            eng.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps): eng.fill_into(out, d_a, d_b)
            e1.record(); torch.cuda.synchronize()
            t = e0.elapsed_time(e1) / reps
            print(x, "s2w", w, "store_policy", pol, "%.3f ms" % t, "%.1f GCUPS" % (n * n / t / 1e6), "strips", eng.get_option("last_strips2"), "tiles", eng.get_option("last_tiles"), flush=True)
    eng.set_option("store_policy", 0); eng.set_option("s2w", 0)
    out.free()
eng.close()
