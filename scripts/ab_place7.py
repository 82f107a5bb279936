# Fill time as a function of the distance P - H inside one arena (streaming stores on both).
import importlib, sys, torch, numpy as np
sys.path.insert(0, '.')
sw = importlib.import_module("smith-waterman_amd")
cols = rows = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
eng = sw.Engine(0)
if len(sys.argv) > 2: eng.set_option("store_policy", int(sys.argv[2]))
a, b = sw.generate(cols, rows, 1); d_a, _ = eng.to_device(a); d_b, _ = eng.to_device(b)
n = (rows + 1) * (cols + 1)
res = torch.zeros(3, dtype=torch.int64, device="cuda")
big = torch.empty(3 * 4 * n + (1100 << 20), dtype=torch.uint8, device="cuda")
def run(offH, offP, reps=5):
    H = big[offH:offH + 4 * n].view(torch.int32).view(rows + 1, cols + 1)
    P = big[offP:offP + 4 * n].view(torch.int32).view(rows + 1, cols + 1)
    out = sw.Fill(H, P, res, cols, rows)
    eng.fill_into(out, d_a, d_b); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): eng.fill_into(out, d_a, d_b)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
base0 = ((4 * n + (2 << 20) - 1) >> 21) << 21
print("arena %x, H at 0, P at %d MB + d" % (big.data_ptr(), base0 >> 20))
M = 1 << 20
print("d = 0..1024 MB step 32 MB:", " ".join("%.2f" % run(0, base0 + k * 32 * M) for k in range(33)))
print("d = 0..32 MB step 1 MB  :", " ".join("%.2f" % run(0, base0 + k * M) for k in range(33)))
print("d = 0..2 MB step 64 KB  :", " ".join("%.2f" % run(0, base0 + k * 65536) for k in range(33)))
print("d = 0..64 KB step 4 KB  :", " ".join("%.2f" % run(0, base0 + k * 4096) for k in range(17)))
print("d = 0..4 KB step 256 B  :", " ".join("%.2f" % run(0, base0 + k * 256) for k in range(17)))
print("H shifted with P (both + x), d = 0: x = 0..2 MB step 128 KB:", " ".join("%.2f" % run(k * 131072, base0 + k * 131072) for k in range(17)))
