import importlib, sys, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from oracle_lib import golden
sw = importlib.import_module("smith-waterman_amd")
eng = sw.Engine(0)
names = ["rand_256x256_s1", "rand_300x200_s1", "rand_129x64_s11"]
gs = {n: golden(n) for n in names}
# warm: size the workspaces
for nm in names: eng.fill(gs[nm]["a"], gs[nm]["b"])
res = {}
for e in range(0, 255):
    for nm in names:
        bad_n = 0
        for rep in range(3):
            eng.set_option("debug_epoch8", e)
            out = eng.fill(gs[nm]["a"], gs[nm]["b"])
            dH = out.H.cpu().numpy()
            if not np.array_equal(dH, gs[nm]["H"]):
                bad = np.argwhere(dH != gs[nm]["H"]); r, c = bad[0]
                bad_n += 1
                if rep == 0 or bad_n == 1:
                    print(f"epoch {e+1} {nm} rep {rep}: {len(bad)} cells first ({r},{c}) got {dH[r,c]}", flush=True)
        if bad_n: res[(e + 1, nm)] = bad_n
print("failing (epoch, problem): count of 3:", res)
