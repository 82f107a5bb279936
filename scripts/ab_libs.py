# A/B of two builds of libswhip.so in ONE process on the SAME output buffers (placement moves a fill by 30 %: different processes cannot
# be compared).  usage: python scripts/ab_libs.py <libA.so> <libB.so> [more .so ...] [cols] [rows]
import ctypes, importlib, sys, time, torch
sys.path.insert(0, '.')
sw = importlib.import_module("smith-waterman_amd")
paths = [x for x in sys.argv[1:] if x.endswith(".so")]
nums = [int(x) for x in sys.argv[1:] if not x.endswith(".so")]
cols = nums[0] if nums else 16384
rows = nums[1] if len(nums) > 1 else cols
torch.cuda.init()
a, b = sw.generate(cols, rows, 1)
d_a = torch.zeros(cols + 16, dtype=torch.uint8, device="cuda"); d_a[:cols] = torch.from_numpy(a)
d_b = torch.zeros(rows + 16, dtype=torch.uint8, device="cuda"); d_b[:rows] = torch.from_numpy(b)
torch.cuda.synchronize()
sc = sw._Scores(3, -3, -2)
libs = []
for path in paths:
    L = ctypes.CDLL(path)
    L.sw_last_error.restype = ctypes.c_char_p
    h = ctypes.c_void_p()
    assert L.sw_create(0, ctypes.byref(h)) == 0
    libs.append((path, L, h))
# buffers from the first library's allocator
L0, h0 = libs[0][1], libs[0][2]
dH, dP = ctypes.c_void_p(), ctypes.c_void_p()
ms = (ctypes.c_float * 16)()
t0 = time.perf_counter()
rc = L0.sw_alloc_outputs(h0, ctypes.c_void_p(d_a.data_ptr()), ctypes.c_int64(cols), ctypes.c_void_p(d_b.data_ptr()), ctypes.c_int64(rows), ctypes.byref(sc), 4, 4, 0,
                         ctypes.byref(dH), ctypes.byref(dP), ms)
torch.cuda.synchronize()
print(f"sw_alloc_outputs({libs[0][0]}): rc {rc}, {1e3 * (time.perf_counter() - t0):.1f} ms, per candidate: {[round(x, 3) for x in ms if x > 0]}")
res = torch.zeros(3, dtype=torch.int64, device="cuda")


def fill(L, h, n):
    for _ in range(n):
        rc = L.sw_fill_device(h, ctypes.c_void_p(d_a.data_ptr()), ctypes.c_int64(cols), ctypes.c_void_p(d_b.data_ptr()), ctypes.c_int64(rows), ctypes.byref(sc), dH, 4, dP, None,
                              ctypes.c_void_p(res.data_ptr()), None)
        assert rc == 0, L.sw_last_error()


for _, L, h in libs:
    fill(L, h, 200)
torch.cuda.synchronize()
for rnd in range(4):
    for path, L, h in libs:
        fill(L, h, 3)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fill(L, h, 20); e1.record(); torch.cuda.synchronize()
        print(f"round {rnd} {path}: {e0.elapsed_time(e1) / 20:.4f} ms per fill  result {res.cpu().tolist()}", flush=True)
