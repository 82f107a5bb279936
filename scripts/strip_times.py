import importlib, sys, numpy as np, torch
sys.path.insert(0, '.')
sw = importlib.import_module("smith-waterman_amd")
cols, rows = int(sys.argv[1]), int(sys.argv[2]); flags = int(sys.argv[3]) if len(sys.argv) > 3 else 0
ns = int(sys.argv[4]) if len(sys.argv) > 4 else 2; nc = int(sys.argv[5]) if len(sys.argv) > 5 else 4
pace = int(sys.argv[6]) if len(sys.argv) > 6 else 0
eng = sw.Engine(0); eng.set_option("pace_ps", pace); eng.set_option("debug_flags", flags); eng.set_option("strips_per_group", ns); eng.set_option("consumers", nc)
a, b = sw.generate(cols, rows, 1); d_a, _ = eng.to_device(a); d_b, _ = eng.to_device(b); out = eng.alloc(cols, rows)
S = (cols + 62) // 63
dbg = torch.zeros(6 * S + 64, dtype=torch.int64, device="cuda")
eng.fill_into(out, d_a, d_b); eng.synchronize()
eng.set_option("debug_buf", dbg.data_ptr())
eng.fill_into(out, d_a, d_b); eng.synchronize()
t = dbg.cpu().numpy()[:2 * S].reshape(S, 2).astype(np.float64) * 0.01  # 100 MHz ticks -> us
t -= t[:, 0].min()
print("strip: start_us end_us  (end-to-end gap to previous)")
for s in list(range(min(S, 12))) + list(range(max(12, S - 4), S)):
    print(f"{s:4d}: {t[s,0]:8.2f} {t[s,1]:8.2f}  dEnd={t[s,1]-t[s-1,1] if s else 0:7.2f}")
print("mean dEnd over all hops: %.2f us; last end %.2f us" % (np.diff(t[:, 1]).mean() if S > 1 else 0, t[:, 1].max()))
raw = dbg.cpu().numpy()
t0 = raw[:2 * S].reshape(S, 2)[:, 0].min()
hx = (raw[2 * S + 8: 2 * S + 8 + 2 * ((S + ns - 1) // ns)].reshape(-1, 2).astype(np.float64) - t0) * 0.01
print("row %d:  group: importer-has-it(us)  exporter-stored-it(us)   [import(g+1) - export(g)]" % (rows // 2))
for g in range(min(len(hx), 6)):
    nxt = hx[g + 1, 0] - hx[g, 1] if g + 1 < len(hx) and hx[g, 1] > 0 else float("nan")
    print(f"   {g}: {hx[g,0]:9.2f} {hx[g,1]:9.2f}   {nxt:6.2f}")
eng.set_option("debug_buf", 0)
d = np.diff(t[:, 1])
if S > 4:
    inwg = d[0::ns] if ns == 2 else d
    cross = d[1::ns] if ns == 2 else d
    for nm, v in (("in-WG hops (odd strip after even)", inwg), ("cross-WG hops", cross)):
        print(f"{nm}: n={len(v)} mean {v.mean():.2f} median {np.median(v):.2f} p10 {np.percentile(v,10):.2f} p90 {np.percentile(v,90):.2f} max {v.max():.2f} sum {v.sum():.1f} us")
    x = hx[1:, 0] - hx[:-1, 1]
    x = x[np.isfinite(x) & (hx[:-1, 1] > 0)]
    print(f"export->import at mid row: mean {x.mean():.2f} median {np.median(x):.2f} p90 {np.percentile(x,90):.2f} max {x.max():.2f}")

pl = raw[4 * S + 16: 4 * S + 16 + 2 * S].reshape(S, 2)
print("slow-path polls per strip (ring back-pressure, halo wait): mean %.0f / %.0f; strips 0..7:" % (pl[:, 0].mean(), pl[:, 1].mean()), pl[:8].tolist())
ev, od = pl[0::2], pl[1::2]
print("  even strips mean bp %.0f halo %.0f | odd strips mean bp %.0f halo %.0f" % (ev[:, 0].mean(), ev[:, 1].mean(), od[:, 0].mean(), od[:, 1].mean()))
