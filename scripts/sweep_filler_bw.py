# The fillers behind scouts are paced so that their stores together stay below `filler_bw_gbs`: what does the fill take for
# other values of that estimate?  Same buffers, one process.
import importlib, sys, torch
sys.path.insert(0, ".")
sw = importlib.import_module("smith-waterman_amd")
eng = sw.Engine(0); eng.set_option("placement_budget_ms", 20000)
sizes = [int(x) for x in sys.argv[1:]] or [16384, 24576, 32768]
for n in sizes:
    a, b = sw.generate(n, n, 1); d_a, _ = eng.to_device(a); d_b, _ = eng.to_device(b)
    out, ms = eng.alloc_outputs(d_a, d_b, n, n)
    for bw in (4200, 3900, 3600, 3300, 3000, 2700, 4200):
        eng.set_option("filler_bw_gbs", bw)
        reps = 20 if n < 20000 else 5
        for _ in range(60 if n < 20000 else 4): eng.fill_into(out, d_a, d_b)
        eng.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): eng.fill_into(out, d_a, d_b)
        e1.record(); torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / reps
        print(n, "filler_bw_gbs", bw, "%.3f ms" % t, "%.1f GCUPS" % (n * n / t / 1e6), "tiles", eng.get_option("last_tiles"), flush=True)
    out.free()
eng.close()
