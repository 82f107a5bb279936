# consumer waves per strip of the two-column kernel (the rest of the nine helper waves import), the library's own geometry, same buffers
import importlib, sys, torch
sys.path.insert(0, ".")
sw = importlib.import_module("smith-waterman_amd")
eng = sw.Engine(0); eng.set_option("placement_budget_ms", 20000); eng.set_option("placement_hold_gib", 48)
for x in sys.argv[1:] or ["24576", "32768"]:
    n, h64 = int(x.rstrip("h")), x.endswith("h")
    a, b = sw.generate(n, n, 1); d_a, _ = eng.to_device(a); d_b, _ = eng.to_device(b)
    out, ms = eng.alloc_outputs(d_a, d_b, n, n, torch.int64 if h64 else torch.int32)
    for nc in (0, 4, 5, 6, 7, 0):
        eng.set_option("consumers", nc)
        reps = 10 if n < 30000 else 5
        for _ in range(10 if n < 30000 else 2): eng.fill_into(out, d_a, d_b)
        eng.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): eng.fill_into(out, d_a, d_b)
        e1.record(); torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / reps
        print(x, "consumers", nc, "%.3f ms" % t, "%.1f GCUPS" % (n * n / t / 1e6), "strips", eng.get_option("last_strips2"), flush=True)
    eng.set_option("consumers", 0)
    out.free()
eng.close()
