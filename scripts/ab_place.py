# Does the run-to-run spread of the fill time follow where H and P were placed?  Same process, several
# allocations at different offsets inside one big buffer, time each.
import importlib, sys, torch, numpy as np
sys.path.insert(0, '.')
sw = importlib.import_module("smith-waterman_amd")
cols = rows = 16384
eng = sw.Engine(0)
a, b = sw.generate(cols, rows, 1); d_a, _ = eng.to_device(a); d_b, _ = eng.to_device(b)
n = (rows + 1) * (cols + 1)
big = torch.empty(3 * n * 4 + (64 << 20), dtype=torch.uint8, device="cuda")
def run(offH, offP, reps=10):
    H = big[offH:offH + 4 * n].view(torch.int32).view(rows + 1, cols + 1)
    P = big[offP:offP + 4 * n].view(torch.int32).view(rows + 1, cols + 1)
    out = sw.Fill(H, P, torch.zeros(3, dtype=torch.int64, device="cuda"), cols, rows)
    for _ in range(2): eng.fill_into(out, d_a, d_b)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): eng.fill_into(out, d_a, d_b)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
print("base ptr %x" % big.data_ptr())
szr = ((4 * n + (2 << 20) - 1) >> 21) << 21   # H size rounded up to 2 MB
for dP in [0, 256, 4096, 65536, 1 << 20, 2 << 20, (2 << 20) + 4096, 3 << 20, 16 << 20, (16 << 20) + 65536 * 8]:
    t = run(0, szr + dP)
    print(f"P - H = H_size_rounded + {dP:>9d}: {t:.3f} ms")
for oH in [0, 256, 4096, 65536, 1 << 20]:
    t = run(oH, szr + (32 << 20) + oH)
    print(f"both shifted by {oH:>8d}: {t:.3f} ms")
