# Placement study, round 4: does a FINE offset between H and P inside ONE allocation decide the fast / slow mode of a fill?
#   part 1: one arena (hipMalloc), H at its start, P at H + hbytes (rounded to 2 MiB) + delta for a sweep of deltas
#   part 2: the lottery -- N plain (hipMalloc H, hipMalloc P) pairs, freed in between or kept
#   part 3: arena of 8-GiB blocks (as scripts/ab_arena.py) with delta = 0 and the best fine delta of part 1
# usage: python scripts/placement_fine.py [cols] [arena_GiB]
import importlib, sys, ctypes, json, torch
sys.path.insert(0, '.')
sw = importlib.import_module("smith-waterman_amd")
eng = sw.Engine(0)
L = sw.lib()
cols = rows = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
arena_gib = int(sys.argv[2]) if len(sys.argv) > 2 else 8
a, b = sw.generate(cols, rows, 1); d_a, _ = eng.to_device(a); d_b, _ = eng.to_device(b)
n = (rows + 1) * (cols + 1) * 4
res = torch.zeros(3, dtype=torch.int64, device="cuda")
sc = sw._Scores(3, -3, -2)
st = torch.cuda.current_stream().cuda_stream


def timed(dH, dP, reps=4):
    sw._check(L.sw_fill_device(eng._h, d_a.data_ptr(), cols, d_b.data_ptr(), rows, ctypes.byref(sc), dH, 4, dP, None, res.data_ptr(), st))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        sw._check(L.sw_fill_device(eng._h, d_a.data_ptr(), cols, d_b.data_ptr(), rows, ctypes.byref(sc), dH, 4, dP, None, res.data_ptr(), st))
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def dmalloc(nbytes):
    p = ctypes.c_void_p()
    sw._check(L.sw_device_malloc(eng._h, nbytes, ctypes.byref(p)))
    return p.value


# warm up clocks
tmpH, tmpP = dmalloc(n), dmalloc(n)
for _ in range(300):
    timed(tmpH, tmpP, 1)
print(f"warm pair: {timed(tmpH, tmpP):.3f} ms  H={tmpH:#x} P={tmpP:#x}", flush=True)
L.sw_device_free(eng._h, tmpH); L.sw_device_free(eng._h, tmpP)

out = {}
G = 1 << 30
M2 = 2 << 20
arena = dmalloc(arena_gib * G)
hb = (n + M2 - 1) // M2 * M2
print(f"part 1: arena {arena_gib} GiB at {arena:#x}, matrix {n / G:.3f} GiB", flush=True)
deltas = [0] + [1 << k for k in range(8, 31)] + [3 << 10, 3 << 12, 3 << 16, 3 << 20, 5 << 20, 7 << 20, (1 << 20) + (1 << 12), (2 << 20) + 256, (1 << 21) + (1 << 16)]
p1 = {}
for d in deltas:
    if hb + d + n > arena_gib * G:
        continue
    t = timed(arena, arena + hb + d)
    p1[d] = t
    print(f"  delta {d:>12d} ({d:#x}): {t:.3f} ms", flush=True)
out["part1"] = p1
# also H not at the arena's start: shift both by k * 64 MiB
print("part 1b: H shifted inside the arena, P right behind (+2 MiB)", flush=True)
for k in range(0, 16):
    h = arena + k * (64 << 20)
    if k * (64 << 20) + hb + M2 + n > arena_gib * G:
        break
    print(f"  H +{k * 64} MiB: {timed(h, h + hb + M2):.3f} ms", flush=True)
L.sw_device_free(eng._h, arena)

print("part 2: plain pairs (hipMalloc H, hipMalloc P), all kept until the end", flush=True)
keep = []
p2 = []
for i in range(10):
    h, p = dmalloc(n), dmalloc(n)
    keep += [h, p]
    t = timed(h, p)
    p2.append(t)
    print(f"  pair {i}: H={h:#x} P={p:#x} dist={(p - h) / G:+.3f} GiB: {t:.3f} ms", flush=True)
# cross pairs: H of pair i with P of pair j
print("part 2b: H of pair 0 with every other buffer as P", flush=True)
for j in range(1, len(keep)):
    print(f"  H={keep[0]:#x} P={keep[j]:#x}: {timed(keep[0], keep[j]):.3f} ms", flush=True)
for x in keep:
    L.sw_device_free(eng._h, x)
out["part2"] = p2
print(json.dumps(out))
eng.close()
