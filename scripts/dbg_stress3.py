import importlib, sys, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from oracle_lib import golden
sw = importlib.import_module("smith-waterman_amd")
names = ["rand_256x256_s1", "rand_300x200_s1", "rand_129x64_s11"]
gs = {nm: golden(nm) for nm in names}
def run(label, n, flags=0, **opts):
    eng = sw.Engine(0)
    eng.set_option("debug_flags", flags)
    for k, v in opts.items(): eng.set_option(k, v)
    fails = 0; first = ""
    for it in range(n):
        for nm in names:
            g = gs[nm]
            out = eng.fill(g["a"], g["b"])
            dH = out.H.cpu().numpy()
            if not np.array_equal(dH, g["H"]):
                fails += 1
                if not first:
                    bad = np.argwhere(dH != g["H"]); r, c = bad[0]
                    first = f"fill {it*3+names.index(nm)+1} {nm} ({r},{c}) got {dH[r,c]}"
    print(f"{label}: {fails} failures in {n*3} fills  {first}", flush=True)
    eng.close()
n = int(sys.argv[1])
run("default (perm)", n)
run("default (perm) again", n)
