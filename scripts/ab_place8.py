# separate allocations (as alloc() does) x P shifted by 0..3 MB inside its own (padded) buffer
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
held = []
M = 1 << 20
for c in range(8):
    Hb = torch.empty(4 * n + 8 * M, dtype=torch.uint8, device="cuda")
    Pb = torch.empty(4 * n + 8 * M, dtype=torch.uint8, device="cuda")
    row = []
    for sh in (0, 1, 2, 3, 4, 6):
        H = Hb[:4 * n].view(torch.int32).view(rows + 1, cols + 1)
        P = Pb[sh * M: sh * M + 4 * n].view(torch.int32).view(rows + 1, cols + 1)
        row.append(t(H, P))
    d = Pb.data_ptr() - Hb.data_ptr()
    print(f"cand {c}: H {Hb.data_ptr():x} P-H = {d/M:.0f} MB (mod 4 MB = {(d % (4*M))/M:.0f}):  shift 0,1,2,3,4,6 MB -> " + " ".join("%.3f" % x for x in row))
    held.append((Hb, Pb))
