import importlib, sys, torch, numpy as np
sys.path.insert(0, '.')
sw = importlib.import_module("smith-waterman_amd")
cols = rows = 16384
eng = sw.Engine(0)
a, b = sw.generate(cols, rows, 1); d_a, _ = eng.to_device(a); d_b, _ = eng.to_device(b)
n = (rows + 1) * (cols + 1)
res = torch.zeros(3, dtype=torch.int64, device="cuda")
def t(H, P, reps=5):
    out = sw.Fill(H, P, res, cols, rows)
    eng.fill_into(out, d_a, d_b); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): eng.fill_into(out, d_a, d_b)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
M = 1 << 20; G = 1 << 30
def view(buf, off): return buf[off:off + 4 * n].view(torch.int32).view(rows + 1, cols + 1)
for gb in (16, 3):
    big = torch.empty(gb * G, dtype=torch.uint8, device="cuda")
    base = 2 * G if gb >= 4 else 1536 * M
    print(f"arena {gb} GB {big.data_ptr():x}: H at 0, P at {base>>20} MB + shift 0,1,2,3 MB:", " ".join("%.3f" % t(view(big, 0), view(big, base + s * M)) for s in range(4)))
    if gb >= 16:
        print("   H at 8 GB, P at 12 GB + shift:", " ".join("%.3f" % t(view(big, 8 * G), view(big, 12 * G + s * M)) for s in range(4)))
    # small separate P with the arena H
    Ps = torch.empty(4 * n + 8 * M, dtype=torch.uint8, device="cuda")
    print("   H in arena, P separate + shift:", " ".join("%.3f" % t(view(big, 0), view(Ps, s * M)) for s in range(4)), " (P-H) mod 4MB = %d MB" % (((Ps.data_ptr() - big.data_ptr()) % (4 * M)) >> 20))
    Hs = torch.empty(4 * n + 8 * M, dtype=torch.uint8, device="cuda")
    print("   H separate, P separate + shift:", " ".join("%.3f" % t(view(Hs, 0), view(Ps, s * M)) for s in range(4)))
    del big, Ps, Hs; torch.cuda.empty_cache()
