# Split strips: which share of its strip should a scout write itself, and from which strip on?  Same buffers, one process.
import importlib, sys, torch
sys.path.insert(0, ".")
sw = importlib.import_module("smith-waterman_amd")
eng = sw.Engine(0); eng.set_option("placement_budget_ms", 20000)
sizes = [int(x) for x in sys.argv[1:]] or [16384, 32768]
for n in sizes:
    a, b = sw.generate(n, n, 1); d_a, _ = eng.to_device(a); d_b, _ = eng.to_device(b)
    out, ms = eng.alloc_outputs(d_a, d_b, n, n)
    nblk = (n + 15) // 16
    for frac, frm in ((0, 0), (0.875, 1), (0.75, 1), (0.625, 1), (0.5, 1), (0.4, 1), (0.5, 40), (0.5, 80), (0.75, 60), (0, 0)):
        eng.set_option("split_blk", int(frac * nblk)); eng.set_option("split_from", frm)
        reps = 20 if n < 20000 else 5
        for _ in range(60 if n < 20000 else 4): eng.fill_into(out, d_a, d_b)
        eng.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): eng.fill_into(out, d_a, d_b)
        e1.record(); torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / reps
        print(n, "split_blk", frac, "split_from", frm, "->", eng.get_option("last_split_from"), "%.3f ms" % t, "%.1f GCUPS" % (n * n / t / 1e6), "tiles", eng.get_option("last_tiles"), out.result()["max_pos"], flush=True)
    eng.set_option("split_blk", 0); eng.set_option("split_from", 0)
    out.free()
eng.close()
