# Whole 64-byte lines per store instruction stream at 6.5-7 TB/s when the lines start at lane 0 (scripts/store_pattern_probe.py).  Does it matter
# WHICH lanes hold a line?  448-byte segments (7 lines, 56 lanes) stored by lanes k .. k + 55 for k = 0 .. 8, and with k moving from row to row
# as the windows of csrc/sw_wholeline_consumer.inc do.
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


for seg in (56, 48):
    for nrg in (2, 8):
        for k in range(0, 9):
            if k + seg > 64: continue
            t, g = probe(16384, 65536, seg, nrg, 128 | (k << 8))
            print(f"{seg * 8} B segments of whole lines in lanes {k}..{k + seg - 1}, nrg {nrg}: {g:7.0f} GB/s ({t:.3f} ms)", flush=True)
        t, g = probe(16384, 65536, seg, nrg, 128 | (1 << 12))
        print(f"{seg * 8} B segments of whole lines, first lane moves every second row, nrg {nrg}: {g:7.0f} GB/s ({t:.3f} ms)", flush=True)
# the windows of overlapping strips themselves: segments every 440 bytes of a row with pitch 65540, each storing the 7 or 8 whole lines that begin inside it
for nrg in (2, 8):
    for name, mode in (("round-robin over the XCDs", 8), ("neighbours on one XCD", 8 | 16)):
        for lag in (0, 300, 2350):
            m = mode | (32 if lag else 0) | (lag << 8)
            t, g = probe(16380, 65540, 55, nrg, m)
            print(f"440-byte windows of whole lines, pitch 65540, {name:26s} lag {lag:5d} ns, nrg {nrg}: {g:7.0f} GB/s ({t:.3f} ms)", flush=True)
    t, g = probe(16380, 65540, 63, nrg, 0)
    print(f"504-byte segments, pitch 65540 (the fill's pattern), nrg {nrg}: {g:7.0f} GB/s ({t:.3f} ms)", flush=True)
eng.close()
