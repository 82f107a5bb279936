import importlib, sys, numpy as np, torch
sys.path.insert(0, '.')
sw = importlib.import_module("smith-waterman_amd")
cols, rows = int(sys.argv[1]), int(sys.argv[2]); flags = int(sys.argv[3]) if len(sys.argv) > 3 else 0
ns = int(sys.argv[4]) if len(sys.argv) > 4 else 2; nc = int(sys.argv[5]) if len(sys.argv) > 5 else 4
eng = sw.Engine(0); eng.set_option("debug_flags", flags); eng.set_option("strips_per_group", ns); eng.set_option("consumers", nc)
a, b = sw.generate(cols, rows, 1); d_a, _ = eng.to_device(a); d_b, _ = eng.to_device(b); out = eng.alloc(cols, rows)
S = (cols + 62) // 63
dbg = torch.zeros(2 * S + 8 + 2 * S + 8, dtype=torch.int64, device="cuda")
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
