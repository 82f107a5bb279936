# Placement study, round 4 (2): can a cheap synthetic two-stream store probe (sw_probe_streams, csrc/sw_place.hip) classify a pair of
# buffers the way trial fills do?  Allocates N buffers of 1 GiB, classifies (buffer 0, k) and (buffer 1, k) with real 16384^2 fills and with
# the probe in several geometries; also times hipMalloc / hipFree of large blocks.
import importlib, sys, ctypes, time, torch
sys.path.insert(0, '.')
sw = importlib.import_module("smith-waterman_amd")
eng = sw.Engine(0)
L = sw.lib()
L.sw_probe_streams.restype = ctypes.c_int
L.sw_probe_streams.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                               ctypes.POINTER(ctypes.c_float)]
cols = rows = 16384
a, b = sw.generate(cols, rows, 1); d_a, _ = eng.to_device(a); d_b, _ = eng.to_device(b)
n = (rows + 1) * (cols + 1) * 4
res = torch.zeros(3, dtype=torch.int64, device="cuda")
sc = sw._Scores(3, -3, -2)
st = torch.cuda.current_stream().cuda_stream
G = 1 << 30


def fill_ms(dH, dP, reps=3):
    sw._check(L.sw_fill_device(eng._h, d_a.data_ptr(), cols, d_b.data_ptr(), rows, ctypes.byref(sc), dH, 4, dP, None, res.data_ptr(), st))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        sw._check(L.sw_fill_device(eng._h, d_a.data_ptr(), cols, d_b.data_ptr(), rows, ctypes.byref(sc), dH, 4, dP, None, res.data_ptr(), st))
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def probe(dX, dY, prows, pitch, seg, nrg, mode, reps=3):
    ms = ctypes.c_float()
    sw._check(L.sw_probe_streams(eng._h, dX, dY, prows, pitch, seg, nrg, mode, reps, ctypes.byref(ms)))
    return ms.value


def dmalloc(nbytes):
    p = ctypes.c_void_p()
    sw._check(L.sw_device_malloc(eng._h, nbytes, ctypes.byref(p)))
    return p.value


for gib in (1, 8, 32, 64):
    torch.cuda.synchronize()
    t0 = time.perf_counter(); p = dmalloc(gib * G); t1 = time.perf_counter()
    L.sw_device_free(eng._h, p); t2 = time.perf_counter()
    print(f"hipMalloc {gib} GiB: {1e3 * (t1 - t0):.2f} ms, hipFree {1e3 * (t2 - t1):.2f} ms", flush=True)

N = int(sys.argv[1]) if len(sys.argv) > 1 else 24
bufs = [dmalloc(n) for _ in range(N)]
for _ in range(300):   # clocks
    fill_ms(bufs[0], bufs[1], 1)
pitch = (cols + 1) * 4
geos = [("fill-like 504B x130 segs, 2 row groups", rows + 1, pitch, 63, 2), ("quarter rows", (rows + 1) // 4, pitch, 63, 2),
        ("512B segs pitch 64KiB x128, 2 rg", 16384, 65536, 64, 2), ("1 row group", rows + 1, pitch, 63, 1), ("4 row groups", rows + 1, pitch, 63, 4)]
for ref in (0, 1, 12):
    if ref >= N: break
    print(f"--- reference buffer {ref} ({bufs[ref]:#x})", flush=True)
    print("   k   fill(ref=H,k=P)  fill(k=H,ref=P) | probes two-stream: " + " | ".join(g[0] for g in geos), flush=True)
    for k in range(N):
        if k == ref: continue
        f1 = fill_ms(bufs[ref], bufs[k]); f2 = fill_ms(bufs[k], bufs[ref])
        pr = [probe(bufs[ref], bufs[k], g[1], g[2], g[3], g[4], 0) for g in geos]
        print(f"  {k:2d}   {f1:.3f}   {f2:.3f} | " + "  ".join(f"{x:.4f}" for x in pr), flush=True)
    one = [probe(bufs[ref], None, g[1], g[2], g[3], g[4], 1) for g in geos]
    two = [probe(bufs[ref], None, g[1], g[2], g[3], g[4], 2) for g in geos]
    print("  one stream (X only): " + "  ".join(f"{x:.4f}" for x in one))
    print("  two streams in one buffer: " + "  ".join(f"{x:.4f}" for x in two), flush=True)
eng.close()
