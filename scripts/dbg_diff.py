import importlib, sys, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import oracle_lib
sw = importlib.import_module("smith-waterman_amd")
orc = oracle_lib.Oracle()
eng = sw.Engine(0)
cols, rows = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(cols * 7919 + rows)
a = rng.integers(0, 4, cols).astype(np.uint8) + 65
b = rng.integers(0, 4, rows).astype(np.uint8) + 65
H, P, mp = orc.fill(a, b)
for rep in range(3):
    out = eng.fill(a, b)
    dH = out.H.cpu().numpy()
    bad = np.argwhere(dH != H)
    print("rep", rep, "mismatches:", len(bad))
    if len(bad):
        print(" first:", bad[:5].tolist(), "cols by strip:", sorted(set((bad[:, 1] - 1) // 63)), "rows range", bad[:, 0].min(), bad[:, 0].max())
        r, c = bad[0]
        print(" got", dH[r, max(0, c - 3):c + 4], "want", H[r, max(0, c - 3):c + 4])
