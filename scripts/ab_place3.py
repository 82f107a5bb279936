import importlib, sys, torch, numpy as np
sys.path.insert(0, '.')
sw = importlib.import_module("smith-waterman_amd")
cols = rows = 16384
eng = sw.Engine(0)
a, b = sw.generate(cols, rows, 1); d_a, _ = eng.to_device(a); d_b, _ = eng.to_device(b)
def ev(fn, reps):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
keep = []
for i in range(8):
    out = eng.alloc(cols, rows)
    t = ev(lambda: eng.fill_into(out, d_a, d_b), 10)
    def probe(M):
        for c in range(0, 16000, 400): M[:, c:c + 1].fill_(1)
    pH = ev(lambda: probe(out.H), 3); pP = ev(lambda: probe(out.P), 3)
    bw = ev(lambda: out.H.fill_(0), 3)
    print(f"alloc {i}: H {out.H.data_ptr():x} P {out.P.data_ptr():x}  fill {t:.3f} ms | column-store probe H {pH*1000:.0f} us P {pP*1000:.0f} us | memset H {bw:.3f} ms")
    if i % 3 == 0: keep.append(out)
    else:
        del out; torch.cuda.empty_cache()
