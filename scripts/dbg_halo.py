import importlib, sys, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from oracle_lib import golden
sw = importlib.import_module("smith-waterman_amd")
eng = sw.Engine(0)
names = ["rand_256x256_s1", "rand_300x200_s1", "rand_129x64_s11"]
gs = {n: golden(n) for n in names}
N = 1024 + 1024 * 8
dbgs = torch.zeros((1300, N), dtype=torch.int64, device="cuda")
eng.set_option("debug_flags", 512)
k = 0
found = 0
for it in range(400):
    for nm in names:
        g = gs[nm]
        dbg = dbgs[k]; k += 1
        eng.set_option("debug_buf", dbg.data_ptr())
        out = eng.fill(g["a"], g["b"])
        dH = out.H.cpu().numpy()
        if nm == "rand_129x64_s11" and not np.array_equal(dH, g["H"]):
            bad = np.argwhere(dH != g["H"])
            print(f"iter {it} {nm}: {len(bad)} bad cells rows {bad[:,0].min()}..{bad[:,0].max()} cols {bad[:,1].min()}..{bad[:,1].max()}")
            d = dbg.cpu().numpy()
            for s0 in sorted(set(((bad[:, 1] - 1) // 63).tolist())):
                base = 1024 + s0 * 1024
                imp = d[base: base + 256]; ring0 = d[base + 256: base + 512] & 0xffffffff; halo = d[base + 512: base + 768] & 0xffffffff
                diff = [u for u in range(1, 129) if (imp[u] & 0xffffffff) != ring0[u - 1]]
                print(f" strip {s0}: steps where ring lane0 != what importers wrote: {diff[:24]}")
                for u in diff[:8]:
                    print(f"   step {u}: importer(wave {imp[u] >> 32}) wrote {imp[u] & 0xffffffff:#x}, ring lane0 {ring0[u-1]:#x}, halo ring now {halo[u-1]:#x}")
                print("   importer values steps 1..8:", [hex(int(x & 0xffffffff)) for x in imp[1:9]])
            found += 1
    if found >= 2: break
print("done", found)
