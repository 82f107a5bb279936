# duration of strip 0's producer (no left neighbour: it never waits for a halo) for a few problem shapes / debug flags
import importlib, sys, numpy as np, torch
sys.path.insert(0, '.')
sw = importlib.import_module("smith-waterman_amd")
eng = sw.Engine(0)
def run(cols, rows, flags, ns=1, nc=0, imp=0, pol=0):
    eng.set_option("debug_flags", flags); eng.set_option("strips_per_group", ns); eng.set_option("consumers", nc)
    eng.set_option("importers", imp); eng.set_option("store_policy", pol)
    a, b = sw.generate(cols, rows, 1); d_a, _ = eng.to_device(a); d_b, _ = eng.to_device(b); out = eng.alloc(cols, rows)
    S = (cols + 62) // 63
    dbg = torch.zeros(6 * S + 64, dtype=torch.int64, device="cuda")
    eng.fill_into(out, d_a, d_b); eng.synchronize()
    eng.set_option("debug_buf", dbg.data_ptr())
    eng.fill_into(out, d_a, d_b); eng.synchronize()
    eng.set_option("debug_buf", 0)
    t = dbg.cpu().numpy()[:2 * S].reshape(S, 2).astype(np.float64) * 0.01
    pl = dbg.cpu().numpy()[4 * S + 16: 4 * S + 16 + 2 * S].reshape(S, 2)
    steps = rows + 63 + (S - 1) + 15
    d0 = t[0, 1] - t[0, 0]
    ck = dbg.cpu().numpy()[6 * S + 48: 6 * S + 50]
    mhz = (ck[1] - ck[0]) / d0 if d0 > 0 else 0
    print(f"  strip 0: {(ck[1]-ck[0])/steps:.1f} shader clocks per step at {mhz:.0f} MHz")
    if flags & 64:
        hw = dbg.cpu().numpy()[6 * S + 32: 6 * S + 32 + 16]
        print("  wave: role simd (HW_ID bits 5:4), cu, wave_id:", [(i, int(h >> 32), int((h >> 4) & 3), int((h >> 8) & 15), int(h & 15)) for i, h in enumerate(hw) if h])
    print(f"cols={cols} rows={rows} flags={flags} ns={ns} nc={nc} imp={imp} pol={pol}: strip0 {d0:8.1f} us = {d0*1000/steps:5.2f} ns/step; bp polls {pl[0,0]} halo polls {pl[0,1]}; last end {t[:,1].max()-t[:,0].min():.1f} us", flush=True)
for args in sys.argv[1:]:
    run(*[int(x) for x in args.split(",")])
