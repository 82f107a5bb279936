# sw_alloc_outputs (probe-based) N times in one process: candidates needed, time, and the fill time into each pair it hands out
import importlib, sys, time, torch
sys.path.insert(0, '.')
sw = importlib.import_module("smith-waterman_amd")
eng = sw.Engine(0)
cols = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
rows = int(sys.argv[2]) if len(sys.argv) > 2 else cols
h64 = len(sys.argv) > 3 and sys.argv[3] == "h64"
n = int(sys.argv[4]) if len(sys.argv) > 4 else 10
a, b = sw.generate(cols, rows, 1); d_a, _ = eng.to_device(a); d_b, _ = eng.to_device(b)
hd = torch.int64 if h64 else torch.int32


def fill_ms(out, reps):
    eng.fill_into(out, d_a, d_b); eng.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        eng.fill_into(out, d_a, d_b)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


plain = eng.alloc(cols, rows, hd)
for _ in range(3 if cols > 30000 else 200):
    eng.fill_into(plain, d_a, d_b)
eng.synchronize()
reps = 2 if cols > 30000 else 10
print(f"plain torch pair: {fill_ms(plain, reps):.3f} ms")
keep = []
for i in range(n):
    eng.synchronize()
    t0 = time.perf_counter()
    out, ms = eng.alloc_outputs(d_a, d_b, cols, rows, hd)
    dt = 1e3 * (time.perf_counter() - t0)
    print(f"alloc {i}: {dt:.1f} ms, {len(ms)} candidate(s) {[round(x, 3) for x in ms]} -> fill {fill_ms(out, reps):.3f} ms", flush=True)
    if i % 2 == 0: keep.append(out)      # (keep some: the next search starts from a different state of the heap)
    else: out.free()
eng.close()
