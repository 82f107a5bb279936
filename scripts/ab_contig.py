# Does physically contiguous memory (hipExtMallocWithFlags(hipDeviceMallocContiguous)) put the fill in its fast mode?
import importlib, sys, ctypes, torch, numpy as np
sys.path.insert(0, '.')
sw = importlib.import_module("smith-waterman_amd")
cols = rows = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
eng = sw.Engine(0)
a, b = sw.generate(cols, rows, 1); d_a, _ = eng.to_device(a); d_b, _ = eng.to_device(b)
n = (rows + 1) * (cols + 1)
res = torch.zeros(3, dtype=torch.int64, device="cuda")
hip = ctypes.CDLL("libamdhip64.so")
hip.hipExtMallocWithFlags.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t, ctypes.c_uint]
class Raw:
    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}
def alloc(nbytes, flags):
    p = ctypes.c_void_p()
    rc = hip.hipExtMallocWithFlags(ctypes.byref(p), nbytes, flags)
    if rc != 0: raise RuntimeError(f"hipExtMallocWithFlags rc={rc}")
    return p.value, torch.as_tensor(Raw(p.value, nbytes), device="cuda")
def t(H, P, reps=5):
    out = sw.Fill(H, P, res, cols, rows)
    eng.fill_into(out, d_a, d_b); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): eng.fill_into(out, d_a, d_b)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
M = 1 << 20
keep = []
for flags, name in ((4, "contiguous"), (0, "default"), (4, "contiguous"), (0, "default")):
    for i in range(4):
        try:
            ph, Hb = alloc(4 * n + 8 * M, flags); pp, Pb = alloc(4 * n + 8 * M, flags)
        except RuntimeError as e:
            print(name, "alloc failed:", e); break
        H = Hb[:4 * n].view(torch.int32).view(rows + 1, cols + 1)
        r = []
        for sh in (0, 2):
            off = ((ph + sh * M) - pp) % (4 * M)
            P = Pb[off:off + 4 * n].view(torch.int32).view(rows + 1, cols + 1)
            r.append(t(H, P))
        print(f"{name:10s} {i}: H {ph:x} P {pp:x}  (P-H) mod 4MB = 0: {r[0]:.3f}   = 2MB: {r[1]:.3f}")
        keep.append((Hb, Pb))
