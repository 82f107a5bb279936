import importlib, sys, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from oracle_lib import golden
sw = importlib.import_module("smith-waterman_amd")
eng = sw.Engine(0)
names = sys.argv[1].split(",")
n = int(sys.argv[2])
fails = 0
gs = {nm: golden(nm) for nm in names}
for it in range(n):
    for nm in names:
        g = gs[nm]
        out = eng.fill(g["a"], g["b"])
        dH = out.H.cpu().numpy()
        if not np.array_equal(dH, g["H"]):
            bad = np.argwhere(dH != g["H"])
            r, c = bad[0]
            fails += 1
            if fails <= 6:
                print(f"iter {it} {nm}: {len(bad)} cells; rows {bad[:,0].min()}..{bad[:,0].max()} cols {bad[:,1].min()}..{bad[:,1].max()}; first ({r},{c}) got {dH[r, max(0,c-2):c+3].tolist()} want {g['H'][r, max(0,c-2):c+3].tolist()}", flush=True)
print(f"{fails} failures in {n * len(names)} fills")
