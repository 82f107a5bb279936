import importlib, sys, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from oracle_lib import golden
sw = importlib.import_module("smith-waterman_amd")
names = ["rand_256x256_s1", "rand_300x200_s1", "rand_129x64_s11"]
gs = {nm: golden(nm) for nm in names}
eng = sw.Engine(0)
k = 0
for it in range(int(sys.argv[1])):
    for nm in names:
        g = gs[nm]; k += 1
        out = eng.fill(g["a"], g["b"])
        dH = out.H.cpu().numpy()
        if not np.array_equal(dH, g["H"]):
            bad = np.argwhere(dH != g["H"]); r, c = bad[0]
            G = int(dH[r, c]) + 2 * (int(r) + int(c))
            print(f"fill {k} epoch {((k-1)%255)+1:3d} ({((k-1)%255)+1:#x}) {nm}: first ({r},{c}) strip {(c-1)//63} H {dH[r,c]} G {G:#x}; H ptr {out.H.data_ptr():#x} P ptr {out.P.data_ptr():#x}", flush=True)
