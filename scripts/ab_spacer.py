# H and P a fixed distance apart in allocation order (a spacer allocated between them, freed afterwards)
import importlib, sys, torch, numpy as np
sys.path.insert(0, '.')
sw = importlib.import_module("smith-waterman_amd")
cols = rows = 16384
eng = sw.Engine(0)
a, b = sw.generate(cols, rows, 1); d_a, _ = eng.to_device(a); d_b, _ = eng.to_device(b)
n = (rows + 1) * (cols + 1)
res = torch.zeros(3, dtype=torch.int64, device="cuda")
def t(H, P, reps=4):
    out = sw.Fill(H, P, res, cols, rows)
    eng.fill_into(out, d_a, d_b); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): eng.fill_into(out, d_a, d_b)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
G = 1 << 30
keep = []
for gap_gb in (0, 8, 16, 24, 30, 31, 32, 33, 34, 40, 48, 64):
    r = []
    for rep in range(5):
        H = torch.empty((rows + 1, cols + 1), dtype=torch.int32, device="cuda")
        sp = torch.empty(gap_gb * G, dtype=torch.uint8, device="cuda") if gap_gb else None
        Pb = torch.empty(4 * n + (4 << 20), dtype=torch.uint8, device="cuda")
        off = ((H.data_ptr() + (2 << 20)) - Pb.data_ptr()) % (4 << 20)
        P = Pb[off:off + 4 * n].view(torch.int32).view(rows + 1, cols + 1)
        del sp
        torch.cuda.empty_cache()
        r.append(t(H, P))
        keep.append((H, Pb))   # keep them so the next repetition lands elsewhere
    print(f"spacer {gap_gb:3d} GiB between H and P: " + " ".join("%.3f" % x for x in r), flush=True)
    keep.clear(); torch.cuda.empty_cache()
