# fill time vs the distance between H and P inside ONE big allocation (placement study)
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
def timed(dH, dP, reps=5):
    for _ in range(2):
        sw._check(L.sw_fill_device(eng._h, d_a.data_ptr(), cols, d_b.data_ptr(), rows, ctypes.byref(sc), dH, 4, dP, None, res.data_ptr(), None))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        sw._check(L.sw_fill_device(eng._h, d_a.data_ptr(), cols, d_b.data_ptr(), rows, ctypes.byref(sc), dH, 4, dP, None, res.data_ptr(), torch.cuda.current_stream().cuda_stream))
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for trial in range(2):
    big = ctypes.c_void_p()
    total = 72 * G
    sw._check(L.sw_device_malloc(eng._h, total, ctypes.byref(big)))
    base = big.value
    print(f"trial {trial}: block at {base:#x} (mod 32 GiB = {(base % (32*G))/G:.2f} GiB)")
    for hoff in (0, 16 * G):
        row = []
        for d in (1.002, 1.5, 2, 3, 4, 6, 8, 12, 16, 20, 24, 28, 31, 32, 33, 40, 48):
            off = int(d * G) // (4 << 20) * (4 << 20) + (2 << 20)     # phase 2 MiB
            if hoff + off + n > total: continue
            ms = timed(base + hoff, base + hoff + off)
            row.append(f"{d:g}:{ms:.3f}")
        print(f"  H at +{hoff // G} GiB; P at +d GiB -> ms:  " + "  ".join(row), flush=True)
    sw._check(L.sw_device_free(eng._h, big))
    # shift the next block by allocating a keeper
    keep = ctypes.c_void_p(); sw._check(L.sw_device_malloc(eng._h, 5 * G, ctypes.byref(keep)))
