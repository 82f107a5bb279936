# Partial 64-byte lines halve the store bandwidth (scripts/store_pattern_probe.py).  Can the L2 merge the two halves of a boundary line when the
# neighbouring row segments are written on ONE XCD, write-back, a hand-off apart -- as the fill's strips are?
import importlib, sys, ctypes, torch
sys.path.insert(0, '.')
sw = importlib.import_module("smith-waterman_amd")
eng = sw.Engine(0)
L = sw.lib()
L.sw_probe_streams.restype = ctypes.c_int
L.sw_probe_streams.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                               ctypes.POINTER(ctypes.c_float)]
cols = rows = 16384
a, b = sw.generate(cols, rows, 1); d_a, _ = eng.to_device(a); d_b, _ = eng.to_device(b)
eng.set_option("placement_budget_ms", 20000)
out, ms = eng.alloc_outputs(d_a, d_b, cols, rows)
print("placement ratio", eng.get_option("last_placement_ratio_x1000") / 1000)
for _ in range(200):
    eng.fill_into(out, d_a, d_b)
eng.synchronize()
X, Y = out.H.data_ptr(), out.P.data_ptr()


def probe(prows, pitch, seg, nrg, mode):
    t = ctypes.c_float()
    sw._check(L.sw_probe_streams(eng._h, X, Y, prows, pitch, seg, nrg, mode, 3, ctypes.byref(t)))
    nseg = pitch // (seg * 8)
    return t.value, prows * nseg * seg * 8 * 2 / (t.value * 1e-3) / 1e9


pr, pi, seg = 16380, 65540, 63
for nrg in (2,):
    for name, mode in (("nt, segments dealt round-robin over the XCDs", 0), ("write-back, round-robin", 4), ("nt, neighbours on one XCD", 16), ("write-back, neighbours on one XCD", 4 | 16),
                       ("nt interior + write-back edge lanes, neighbours on one XCD", 64 | 16), ("nt interior + write-back edges, round-robin", 64)):
        for lag in (0, 300, 2350):
            m = mode | (32 if lag else 0) | (lag << 8)
            t, g = probe(pr, pi, seg, nrg, m)
            print(f"{name:62s} lag {lag:5d} ns: {g:7.0f} GB/s ({t:.3f} ms)", flush=True)
t, g = probe(16384, 65536, 64, 2, 0)
print(f"aligned 512-byte segments for comparison: {g:.0f} GB/s")
eng.close()
