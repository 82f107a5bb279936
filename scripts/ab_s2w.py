# strips every 126 columns against overlapping strips (every 110, whole-line stores): the same buffers, one process
import importlib, sys, torch
sys.path.insert(0, ".")
sw = importlib.import_module("smith-waterman_amd")
eng = sw.Engine(0); eng.set_option("placement_budget_ms", 20000)
cases = [(16384, False), (32768, False), (65536, False), (65536, True)] if len(sys.argv) < 2 else [(int(x.rstrip('h')), x.endswith('h')) for x in sys.argv[1:]]
for n, h64 in cases:
    a, b = sw.generate(n, n, 1); d_a, _ = eng.to_device(a); d_b, _ = eng.to_device(b)
    out, ms = eng.alloc_outputs(d_a, d_b, n, n, torch.int64 if h64 else torch.int32)
    for w in (0, 110):
        eng.set_option("s2w", w)
        reps = 20 if n < 30000 else 3
        for _ in range(100 if n < 30000 else 1): eng.fill_into(out, d_a, d_b)
        eng.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): eng.fill_into(out, d_a, d_b)
        e1.record(); torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / reps
        bpc = 12 if h64 else 8
        print("s2w", w, n, "h64" if h64 else "i32", "%.3f ms" % t, "%.1f GCUPS" % (n * n / t / 1e6), "%.0f GB/s" % (bpc * n * n / t / 1e6), "strips", eng.get_option("last_strips2"),
              "tiles", eng.get_option("last_tiles"), out.result(), flush=True)
    eng.set_option("s2w", 0)
    out.free()
eng.close()
