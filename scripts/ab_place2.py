import importlib, sys, torch, numpy as np
sys.path.insert(0, '.')
sw = importlib.import_module("smith-waterman_amd")
cols = rows = 16384
eng = sw.Engine(0)
a, b = sw.generate(cols, rows, 1); d_a, _ = eng.to_device(a); d_b, _ = eng.to_device(b)
def timeit(out, reps=10):
    for _ in range(2): eng.fill_into(out, d_a, d_b)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): eng.fill_into(out, d_a, d_b)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
keep = []
for i in range(6):
    out = eng.alloc(cols, rows)
    t = [timeit(out) for _ in range(3)]
    print(f"alloc {i}: H {out.H.data_ptr():x} P {out.P.data_ptr():x}  {t[0]:.3f} {t[1]:.3f} {t[2]:.3f} ms")
    if i % 2 == 0: keep.append(out)   # hold some so later allocations land elsewhere
    else:
        del out; torch.cuda.empty_cache()
