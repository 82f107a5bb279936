# per-(strip, 16-row block) completion stamps: where and when do consumers (and so the whole chain) stall?
import importlib, sys, torch, numpy as np
sys.path.insert(0, '.')
sw = importlib.import_module("smith-waterman_amd")
cols = rows = 16384
eng = sw.Engine(0)
a, b = sw.generate(cols, rows, 1); d_a, _ = eng.to_device(a); d_b, _ = eng.to_device(b)
S = (cols + 62) // 63; NB = (rows + 15) // 16
dbg = torch.zeros(6 * S + 64 + S * NB, dtype=torch.int64, device="cuda")
for i in range(1):
    out = eng.alloc(cols, rows)
    for _ in range(3): eng.fill_into(out, d_a, d_b)
    eng.synchronize()
    eng.set_option("debug_flags", 128); eng.set_option("debug_buf", dbg.data_ptr()); eng.fill_into(out, d_a, d_b); eng.synchronize()
    eng.set_option("debug_buf", 0); eng.set_option("debug_flags", 0)
    raw = dbg.cpu().numpy()
    t0 = raw[:2 * S].reshape(S, 2)[:, 0].min()
    tend = (raw[:2 * S].reshape(S, 2)[:, 1].max() - t0) * 0.01
    bt = (raw[6 * S + 64:].reshape(S, NB).astype(np.float64) - t0) * 0.01   # us
    print(f"alloc {i}: total {tend:.0f} us")
    # lag of strip s+1 behind strip s at the same block, as a function of the block index
    lag = bt[1:] - bt[:-1]                      # (S-1, NB)
    growth = lag[:, -1] - lag[:, 8]             # how much each hop's lag grew between block 8 and the end
    print("  lag at block 8: mean %.2f  at last block: mean %.2f  (sum of growth %.0f us)" % (lag[:, 8].mean(), lag[:, -1].mean(), growth.sum()))
    # where does the lag grow?  jumps of a hop's lag between consecutive blocks
    jump = np.diff(lag, axis=1)
    big = np.argwhere(jump > 2.0)
    print("  lag jumps > 2 us: %d; total %.0f us" % (len(big), jump[jump > 2.0].sum()))
    if len(big):
        rowsb = big[:, 1]; strips = big[:, 0] + 1; times = bt[strips, rowsb + 1]
        print("   by block index (quartiles):", np.percentile(rowsb, [0, 25, 50, 75, 100]).astype(int).tolist())
        print("   by strip (quartiles):", np.percentile(strips, [0, 25, 50, 75, 100]).astype(int).tolist())
        print("   by time us (quartiles):", np.percentile(times, [0, 25, 50, 75, 100]).astype(int).tolist())
        h, _ = np.histogram(rowsb, bins=16, range=(0, NB)); print("   histogram over block index:", h.tolist())
        h, _ = np.histogram(times, bins=16, range=(0, tend)); print("   histogram over time:", h.tolist())
        h, _ = np.histogram(strips % 8, bins=8, range=(0, 8)); print("   by strip%8:", h.tolist())
        h, _ = np.histogram((strips // 2) % 8, bins=8, range=(0, 8)); print("   by group%8 (XCD):", h.tolist())
    del out; torch.cuda.empty_cache()
    if i == 0:
        for sA in (40, 41, 120, 200):
            l = bt[sA + 1] - bt[sA]
            print(f"  lag strip {sA+1} vs {sA} every 64 blocks:", np.round(l[::64], 1).tolist())
        sp = np.diff(bt[100][::64]) / 64   # us per block
        print("  strip 100: us per block over time:", np.round(sp, 3).tolist())
        print("  strip 100 first block done at %.0f us, strip 101 %.0f, strip 102 %.0f" % (bt[100, 0], bt[101, 0], bt[102, 0]))
        l0 = bt[1:, 0] - bt[:-1, 0]; lE = bt[1:, -1] - bt[:-1, -1]
        print("  lag at block 0: in-WG mean %.2f cross mean %.2f | at end: in-WG %.2f cross %.2f" % (l0[0::2].mean(), l0[1::2].mean(), lE[0::2].mean(), lE[1::2].mean()))
