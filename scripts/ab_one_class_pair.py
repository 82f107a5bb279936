# A pair of output buffers in ONE class of the HBM (the usual outcome of two plain allocations: sw_alloc_outputs with trials = 1): the library's
# choice for it (overlapping strips, streamed whole lines) against 126-column strips, same buffers
import importlib, sys, torch
sys.path.insert(0, ".")
sw = importlib.import_module("smith-waterman_amd")
eng = sw.Engine(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
a, b = sw.generate(n, n, 1); d_a, _ = eng.to_device(a); d_b, _ = eng.to_device(b)
keep = []
for attempt in range(8):
    out, _ = eng.alloc_outputs(d_a, d_b, n, n, trials=1)
    ratio = eng.get_option("last_placement_ratio_x1000") / 1000
    res = []
    for w in (0, 126, 0, 126):
        eng.set_option("s2w", w)
        for _ in range(40): eng.fill_into(out, d_a, d_b)
        eng.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): eng.fill_into(out, d_a, d_b)
        e1.record(); torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / 20
        res.append(f"s2w {w}: {t:.3f} ms {n * n / t / 1e6:.1f} GCUPS ({eng.get_option('last_strips2')} strips)")
    eng.set_option("s2w", 0)
    print(f"plain pair {attempt}: probe ratio {ratio:.3f} | " + " | ".join(res), flush=True)
    keep.append(out)           # (keep it: the next plain pair lands elsewhere)
eng.close()
