# Placement study, round 4 (3): buffers spread over the HBM with (temporary) spacer allocations between them, the full pairwise matrix of
# real 16384^2 fill times, and the synthetic two-stream probe on the same pairs -- does the probe see what the fill sees?
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
    rc = L.sw_device_malloc(eng._h, nbytes, ctypes.byref(p))
    return p.value if rc == 0 else None


gaps = [0, 0, 30, 30, 30, 30, 30, 30, 30, 30]      # GiB of spacer in front of buffer k
bufs, spacers, pos = [], [], 0
for g in gaps:
    if g:
        s = dmalloc(g * G)
        if s is None: break
        spacers.append(s); pos += g
    p = dmalloc(n)
    if p is None: break
    bufs.append((p, pos)); pos += 1
print("buffers (address, GiB allocated before it):", [(hex(p), o) for p, o in bufs], flush=True)
for _ in range(300):
    fill_ms(bufs[0][0], bufs[1][0], 1)
N = len(bufs)
print("pairwise fill ms (row: H, column: P)")
M = [[0.0] * N for _ in range(N)]
for i in range(N):
    for j in range(N):
        if i != j: M[i][j] = fill_ms(bufs[i][0], bufs[j][0])
    print(f"  {bufs[i][1]:4d}: " + " ".join(f"{x:6.3f}" if x else "   -  " for x in M[i]), flush=True)
pitch = (cols + 1) * 4
geos = [("504B nt 2rg", rows + 1, pitch, 63, 2, 0), ("504B nt 8rg", rows + 1, pitch, 63, 8, 0), ("quarter rows 4rg", (rows + 1) // 4, pitch, 63, 4, 0),
        ("504B wb 2rg", rows + 1, pitch, 63, 2, 4), ("504B wb 8rg", rows + 1, pitch, 63, 8, 4), ("contig 2rg", 16384, 65536, 64, 4, 0)]
for name, pr, pi, seg, nrg, md in geos:
    print(f"probe '{name}' ms (row: X, column: Y)")
    for i in range(N):
        row = []
        for j in range(N):
            row.append(f"{probe(bufs[i][0], bufs[j][0], pr, pi, seg, nrg, md):6.3f}" if i != j else "   -  ")
        print(f"  {bufs[i][1]:4d}: " + " ".join(row), flush=True)
eng.close()
