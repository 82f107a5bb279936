# Beyond the reach of the scouts: column tiles (126-column strips, scouts in every tile) against ONE launch of overlapping strips with streaming
# whole-line stores.  Same buffers, one process.
import importlib, sys, torch
sys.path.insert(0, ".")
sw = importlib.import_module("smith-waterman_amd")
eng = sw.Engine(0); eng.set_option("placement_budget_ms", 20000)
for x in sys.argv[1:] or ["24576", "32768"]:
    n, h64 = int(x.rstrip("h")), x.endswith("h")
    a, b = sw.generate(n, n, 1); d_a, _ = eng.to_device(a); d_b, _ = eng.to_device(b)
    out, ms = eng.alloc_outputs(d_a, d_b, n, n, torch.int64 if h64 else torch.int32)
    for name, w, pol, dbg in (("library's choice", 0, 0, 0), ("untiled, strips every 110, streaming", 110, 2, 524288), ("untiled, strips every 126, streaming", 126, 2, 524288),
                              ("tiles allowed, strips every 110, streaming", 110, 2, 0)):
        eng.set_option("store_policy", pol); eng.set_option("s2w", w); eng.set_option("debug_flags", dbg)
        reps = 20 if n < 30000 else 5
        for _ in range(40 if n < 30000 else 2): eng.fill_into(out, d_a, d_b)
        eng.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): eng.fill_into(out, d_a, d_b)
        e1.record(); torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / reps
        print(x, f"{name:40s}", "%.3f ms" % t, "%.1f GCUPS" % (n * n / t / 1e6), "strips", eng.get_option("last_strips2"), "tiles", eng.get_option("last_tiles"), "scouts", eng.get_option("last_scouts"), flush=True)
    eng.set_option("store_policy", 0); eng.set_option("s2w", 0); eng.set_option("debug_flags", 0)
    out.free()
eng.close()
