# Both matrices carved from one big arena: does the arena size decide whether streaming stores are fast?
import importlib, sys, torch, numpy as np
sys.path.insert(0, '.')
sw = importlib.import_module("smith-waterman_amd")
cols = rows = 16384
eng = sw.Engine(0)
a, b = sw.generate(cols, rows, 1); d_a, _ = eng.to_device(a); d_b, _ = eng.to_device(b)
n = (rows + 1) * (cols + 1)
res = torch.zeros(3, dtype=torch.int64, device="cuda")
def run(big, offH, offP, reps=6):
    H = big[offH:offH + 4 * n].view(torch.int32).view(rows + 1, cols + 1)
    P = big[offP:offP + 4 * n].view(torch.int32).view(rows + 1, cols + 1)
    out = sw.Fill(H, P, res, cols, rows)
    eng.fill_into(out, d_a, d_b); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): eng.fill_into(out, d_a, d_b)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for gb in (3, 4, 6, 8, 16, 32):
    big = torch.empty(gb << 30, dtype=torch.uint8, device="cuda")
    g = 1 << 30
    r = [run(big, 0, (gb << 30) - 4 * n - 4096 if gb < 4 else 2 * g), run(big, 0, 1536 << 20) if gb >= 3 else 0]
    if gb >= 8: r += [run(big, 4 * g, 6 * g), run(big, 1 * g, 5 * g)]
    print(f"arena {gb:2d} GB at {big.data_ptr():x}: " + " ".join("%.3f" % x for x in r))
    del big; torch.cuda.empty_cache()
# separate allocations of different sizes for H and P (padding the allocation beyond the matrix)
for pad_gb in (0, 1, 3, 7):
    ts = []
    for rep in range(4):
        Hb = torch.empty(4 * n + (pad_gb << 30), dtype=torch.uint8, device="cuda")
        Pb = torch.empty(4 * n + (pad_gb << 30), dtype=torch.uint8, device="cuda")
        out = sw.Fill(Hb[:4 * n].view(torch.int32).view(rows + 1, cols + 1), Pb[:4 * n].view(torch.int32).view(rows + 1, cols + 1), res, cols, rows)
        eng.fill_into(out, d_a, d_b); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(6): eng.fill_into(out, d_a, d_b)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 6)
        del Hb, Pb, out; torch.cuda.empty_cache()
    print(f"separate allocations padded by {pad_gb} GB: " + " ".join("%.3f" % x for x in ts))
