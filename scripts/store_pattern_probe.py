# What does the HBM write path like?  The two-stream store probe (csrc/sw_place.hip) on an H / P pair in different classes (from
# sw_alloc_outputs) with segment size, alignment and row pitch varied: GB/s per pattern.  (why a fill's stores reach
# ~3.5 TB/s where a contiguous fill_ reaches 6.8.)
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
GiB = 1 << 30


def probe(prows, pitch, seg, nrg, mode, off=0):
    t = ctypes.c_float()
    sw._check(L.sw_probe_streams(eng._h, X + off, Y + off, prows, pitch, seg, nrg, mode, 3, ctypes.byref(t)))
    nseg = pitch // (seg * 8)
    nbytes = prows * nseg * seg * 8 * (1 if mode == 1 else 2)
    return t.value, nbytes / (t.value * 1e-3) / 1e9


cases = [("512 B segments, pitch 65536 (aligned, contiguous rows)", 16384, 65536, 64),
         ("504 B segments, pitch 65536 (rows aligned, segments at multiples of 504)", 16384, 65536, 63),
         ("512 B segments, pitch 65540 (rows drift by 4 B)", 16380, 65540, 64),
         ("504 B segments, pitch 65540 (the fill's pattern)", 16380, 65540, 63),
         ("512 B segments, pitch 65600 (rows drift by 64 B)", 16360, 65600, 64),
         ("512 B segments, pitch 65664 (rows drift by 128 B)", 16340, 65664, 64),
         ("256 B segments, pitch 65536", 16384, 65536, 32),
         ("496 B segments, pitch 65536", 16384, 65536, 62),
         ("480 B segments (a multiple of 32 B), pitch 65536", 16384, 65536, 60)]
for name, pr, pi, seg in cases:
    for nrg in (2, 8):
        t2, g2 = probe(pr, pi, seg, nrg, 0)
        t1, g1 = probe(pr, pi, seg, nrg, 1)
        print(f"{name:75s} nrg {nrg}: two streams {g2:7.0f} GB/s ({t2:.3f} ms), one stream {g1:7.0f} GB/s", flush=True)
# the same fill-like pattern with the row start shifted so that every segment starts on a 64-byte boundary at row 0
for off in (0, 4, 32, 60):
    t2, g2 = probe(16380, 65540, 63, 8, 0, off)
    print(f"fill pattern, base shifted by {off} B: {g2:7.0f} GB/s")
eng.close()
