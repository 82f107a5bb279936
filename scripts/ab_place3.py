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
for i in range(10):
    out = eng.alloc(cols, rows)
    t = ev(lambda: eng.fill_into(out, d_a, d_b), 10)
    p1 = ev(lambda: out.H[:, ::64].fill_(1), 5); p2 = ev(lambda: out.P[:, ::64].fill_(1), 5)
    p3 = ev(lambda: (out.H[:, 5::63].fill_(1), out.P[:, 5::63].fill_(1)), 5)
    idx = torch.randint(0, (rows + 1) * (cols + 1) // 64, (1 << 22,), device="cuda") * 64
    hf = out.H.view(-1)
    p4 = ev(lambda: hf.index_fill_(0, idx, 1), 5)
    print(f"alloc {i}: H {out.H.data_ptr():x} P {out.P.data_ptr():x}  fill {t:.3f} ms | strided-store probes H {p1*1000:.0f} us P {p2*1000:.0f} us both {p3*1000:.0f} us | random 4M stores {p4*1000:.0f} us")
    if i % 3 == 0: keep.append(out)
    else:
        del out, hf; torch.cuda.empty_cache()
