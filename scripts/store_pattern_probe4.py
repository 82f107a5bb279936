# How much can the workgroups of HALF the CUs store?  The two-stream probe with 130 workgroups (one per row segment: what the fillers of a
# 16384-column fill are), 260, 520 and 1040: 504-byte segments at pitch 65540 (the fill's pattern) and 512-byte aligned ones.
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
    return t.value, prows * nseg * seg * 8 * (1 if mode == 1 else 2) / (t.value * 1e-3) / 1e9, nseg * nrg


for name, pr, pi, seg in (("504-byte segments, pitch 65540", 16380, 65540, 63), ("512-byte segments, pitch 65536", 16384, 65536, 64), ("440-byte windows of whole lines, pitch 65540", 16380, 65540, 55)):
    for nrg in (1, 2, 4, 8):
        for mode, what in ((0, "two streams"), (1, "one stream")):
            m = (8 if seg == 55 else 0) if mode == 0 else mode
            if seg == 55 and mode == 1: continue
            t, g, wgs = probe(pr, pi, seg, nrg, m)
            print(f"{name:46s} {wgs:5d} workgroups of 4 waves, {what:11s}: {g:7.0f} GB/s ({t:.3f} ms)", flush=True)
eng.close()
