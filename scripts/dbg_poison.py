import importlib, sys, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from oracle_lib import golden
sw = importlib.import_module("smith-waterman_amd")
names = ["rand_256x256_s1", "rand_300x200_s1", "rand_129x64_s11"]
gs = {nm: golden(nm) for nm in names}
eng = sw.Engine(0)
eng.set_option("debug_flags", 2048)
k = 0; nf = 0
for it in range(int(sys.argv[1])):
    for nm in names:
        g = gs[nm]; k += 1
        out = eng.fill(g["a"], g["b"])
        dH = out.H.cpu().numpy()
        if not np.array_equal(dH, g["H"]):
            nf += 1
            if nf <= 12:
                bad = np.argwhere(dH != g["H"]); r, c = bad[0]
                G = int(dH[r, c]) + 2 * (int(r) + int(c))
                print(f"fill {k} epoch {((k-1)%255)+1} {nm}: first ({r},{c}) strip {(c-1)//63} H {dH[r,c]} G {G:#x}  (poison payload 0x700000+slot -> slot {G - 0x700000 if 0x700000 <= G < 0x700400 else 'n/a'})", flush=True)
print(nf, "failures in", k)
