import importlib, sys, torch, numpy as np
sys.path.insert(0, '.')
sw = importlib.import_module("smith-waterman_amd")
cols = rows = 16384
eng = sw.Engine(0)
a, b = sw.generate(cols, rows, 1); d_a, _ = eng.to_device(a); d_b, _ = eng.to_device(b)
S = (cols + 62) // 63
dbg = torch.zeros(6 * S + 64, dtype=torch.int64, device="cuda")
def ev(fn, reps):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for i in range(8):
    out = eng.alloc(cols, rows)
    t = ev(lambda: eng.fill_into(out, d_a, d_b), 10)
    eng.set_option("debug_buf", dbg.data_ptr()); eng.fill_into(out, d_a, d_b); eng.synchronize(); eng.set_option("debug_buf", 0)
    raw = dbg.cpu().numpy()
    tt = raw[:2 * S].reshape(S, 2).astype(np.float64) * 0.01; tt -= tt[:, 0].min()
    d = np.diff(tt[:, 1]); inwg, cross = d[0::2], d[1::2]
    pl = raw[4 * S + 16: 4 * S + 16 + 2 * S].reshape(S, 2)
    top = np.sort(d)[-8:][::-1]
    topi = (np.argsort(d)[-12:][::-1] + 1).tolist()
    print(f"alloc {i}: fill {t:.3f} ms | strip0 {tt[0,1]:.0f} us | in-WG mean {inwg.mean():.2f} med {np.median(inwg):.2f} | cross mean {cross.mean():.2f} med {np.median(cross):.2f} | bp polls {pl[:,0].mean():.0f} | top hops {np.round(top,1).tolist()} at strips {topi}")
    del out; torch.cuda.empty_cache()
