# Scan the GPU's memory: allocate H/P pairs one after the other (keeping them all) and time a few fills into each.
import importlib, sys, torch, numpy as np
sys.path.insert(0, '.')
sw = importlib.import_module("smith-waterman_amd")
cols = rows = 16384
eng = sw.Engine(0)
a, b = sw.generate(cols, rows, 1); d_a, _ = eng.to_device(a); d_b, _ = eng.to_device(b)
def t(out, reps=4):
    eng.fill_into(out, d_a, d_b); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): eng.fill_into(out, d_a, d_b)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
outs, ts = [], []
for k in range(int(sys.argv[1]) if len(sys.argv) > 1 else 100):
    try:
        o = eng.alloc(cols, rows)
    except RuntimeError:
        break
    outs.append(o); ts.append(t(o))
    if len(ts) % 10 == 0:
        print(" ".join("%.2f" % x for x in ts[-10:]), flush=True)
print(" ".join("%.2f" % x for x in ts[len(ts) - len(ts) % 10:]))
ts = np.array(ts)
print(f"{len(ts)} pairs ({len(ts) * 2.15:.0f} GB): fast (<1.36 ms) {np.sum(ts < 1.36)}, mid {np.sum((ts >= 1.36) & (ts < 1.46))}, slow {np.sum(ts >= 1.46)}")
