import importlib, sys, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from oracle_lib import golden
sw = importlib.import_module("smith-waterman_amd")
eng = sw.Engine(0)
names = sys.argv[1].split(","); n = int(sys.argv[2]); flags = int(sys.argv[3])
eng.set_option("debug_flags", flags)
fails = {}
gs = {nm: golden(nm) for nm in names}
k = 0
for it in range(n):
    for nm in names:
        g = gs[nm]; k += 1
        out = eng.fill(g["a"], g["b"])
        dH = out.H.cpu().numpy()
        if not np.array_equal(dH, g["H"]):
            bad = np.argwhere(dH != g["H"])
            key = (nm, k % 3 if flags & 1024 else k % 255)
            fails[key] = fails.get(key, 0) + 1
            if sum(fails.values()) <= 4:
                r, c = bad[0]
                print(f"fill {k} {nm}: {len(bad)} cells; rows {bad[:,0].min()}..{bad[:,0].max()} cols {bad[:,1].min()}..{bad[:,1].max()}; first ({r},{c}) got {dH[r, max(0,c-2):c+3].tolist()}", flush=True)
print(f"flags {flags}: {sum(fails.values())} failures in {k} fills:", fails)
