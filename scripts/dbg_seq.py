import importlib, sys, numpy as np, glob, os
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import oracle_lib
from oracle_lib import golden, GOLDEN
sw = importlib.import_module("smith-waterman_amd")
eng = sw.Engine(0)
flags = int(sys.argv[1]) if len(sys.argv) > 1 else 0
eng.set_option("debug_flags", flags)
names = sorted(os.path.basename(f)[:-4] for f in glob.glob(os.path.join(GOLDEN, "*.npz")))
for rep in range(3):
    for name in names:
        g = golden(name)
        out = eng.fill(g["a"], g["b"])
        dH, dP = out.H.cpu().numpy(), out.P.cpu().numpy()
        for nm, got, want in (("H", dH, g["H"]), ("P", dP, g["P0"])):
            bad = np.argwhere(got != want)
            if len(bad):
                r, c = bad[0]
                print(f"rep {rep} {name} {nm}: {len(bad)} cells differ; rows {bad[:,0].min()}..{bad[:,0].max()} cols {bad[:,1].min()}..{bad[:,1].max()}; first ({r},{c}) got {got[r, max(0,c-2):c+3].tolist()} want {want[r, max(0,c-2):c+3].tolist()}")
                rows_bad = sorted(set(bad[:, 0].tolist()))
                print("   rows:", rows_bad[:40])
print("done")
