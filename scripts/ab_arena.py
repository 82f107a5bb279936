# fill time for H and P at a grid of offsets inside ONE big allocation (placement study)
import importlib, sys, ctypes, numpy as np, torch
sys.path.insert(0, '.')
sw = importlib.import_module("smith-waterman_amd")
eng = sw.Engine(0)
L = sw.lib()
cols = rows = 16384
a, b = sw.generate(cols, rows, 1); d_a, _ = eng.to_device(a); d_b, _ = eng.to_device(b)
n = (rows + 1) * (cols + 1) * 4
G = 1 << 30
res = torch.zeros(3, dtype=torch.int64, device="cuda")
sc = sw._Scores(3, -3, -2)
def timed(dH, dP, reps=3):
    sw._check(L.sw_fill_device(eng._h, d_a.data_ptr(), cols, d_b.data_ptr(), rows, ctypes.byref(sc), dH, 4, dP, None, res.data_ptr(), None))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        sw._check(L.sw_fill_device(eng._h, d_a.data_ptr(), cols, d_b.data_ptr(), rows, ctypes.byref(sc), dH, 4, dP, None, res.data_ptr(), torch.cuda.current_stream().cuda_stream))
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
total = int(sys.argv[1]) * G
big = ctypes.c_void_p()
sw._check(L.sw_device_malloc(eng._h, total, ctypes.byref(big)))
base = big.value
step = int(sys.argv[2]) if len(sys.argv) > 2 else 8
offs = list(range(0, total // G - 2, step))
print(f"arena {total // G} GiB at {base:#x}; rows: H offset (GiB), columns: P offset (GiB) (+2 MiB phase)")
print("      " + " ".join(f"{p:6d}" for p in offs))
for h in offs:
    row = []
    for p in offs:
        if abs(p - h) < 2: row.append("   -  "); continue
        row.append(f"{timed(base + h * G, base + p * G + (2 << 20)):6.3f}")
    print(f"{h:4d}: " + " ".join(row), flush=True)
