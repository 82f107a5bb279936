# sw_fill_host (host buffers in, host buffers out: INTEGRATION.md section 2) end to end: wall time of the call, PCIe included --
# into matrices fresh from calloc (what the reference's main hands over: the first writer pays the page faults) and into touched ones
import ctypes, importlib, sys, time, numpy as np
sys.path.insert(0, ".")
sw = importlib.import_module("smith-waterman_amd")
eng = sw.Engine(0)
L = sw.lib()
for n in [int(x) for x in sys.argv[1:]] or [16384]:
    a, b = sw.generate(n, n, 1)
    sc, r = sw._Scores(3, -3, -2), sw._Result()
    def call(H, P):
        t0 = time.perf_counter()
        sw._check(L.sw_fill_host(eng._h, a.ctypes.data, n, b.ctypes.data, n, ctypes.byref(sc), H.ctypes.data, P.ctypes.data, ctypes.byref(r)))
        return time.perf_counter() - t0
    call(np.zeros((n + 1, n + 1), np.int32), np.zeros((n + 1, n + 1), np.int32))            # (workspaces, clocks)
    fresh = min(call(np.zeros((n + 1, n + 1), np.int32), np.zeros((n + 1, n + 1), np.int32)) for _ in range(3))
    H, P = np.ones((n + 1, n + 1), np.int32), np.ones((n + 1, n + 1), np.int32)
    touched = min(call(H, P) for _ in range(3))
    gb = 2 * (n + 1) * (n + 1) * 4 / 1e9
    print(f"{n}x{n}: sw_fill_host into fresh calloc'ed matrices {fresh * 1e3:.1f} ms, into touched ones {touched * 1e3:.1f} ms "
          f"({gb:.2f} GB of H + P: {gb / touched:.1f} GB/s host-visible, {n * n / touched / 1e9:.2f} GCUPS host to host); max_pos {r.max_pos}", flush=True)
eng.close()
