import importlib, sys, os
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
sw = importlib.import_module("smith-waterman_amd")
import oracle_lib
orc = oracle_lib.Oracle()
eng = sw.Engine(0)
cols, rows = 200, 100
a, b = orc.generate(cols, rows, 41)
H, P, mp = orc.fill(a, b)
S = (cols + 62) // 63
for flags in (512, 512 | 4096, 512 | 8192):
    dbg = torch.zeros(1024 + S * 1024 + 1024, dtype=torch.int64, device="cuda")
    eng.set_option("debug_flags", flags); eng.set_option("debug_buf", dbg.data_ptr())
    out = eng.fill(a, b); eng.synchronize()
    eng.set_option("debug_buf", 0); eng.set_option("debug_flags", 0)
    bad = np.argwhere(out.H.cpu().numpy() != H)
    print("flags", flags, "bad cells", len(bad), "first", bad[0] if len(bad) else None)
    d = dbg.cpu().numpy()
    for s0 in range(1, S):
        base = 1024 + s0 * 1024
        imp = d[base + 1: base + 200]                      # importer writes per step u (wave<<32 | value)
        used = d[base + 256: base + 256 + 199] & 0xffffffff       # ring lane 0, steps 1.. = the halo values the producer consumed
        halo = d[base + 512: base + 512 + 199] & 0xffffffff       # final halo ring
        diff = np.nonzero(used != halo)[0]
        print(f"  strip {s0}: steps where consumed halo != final halo ring: {list(diff[:20] + 1)}")
        for u in diff[:6]:
            print(f"     step {u+1}: consumed {int(used[u]):#x} final {int(halo[u]):#x} importer-wave {int(imp[u] >> 32)}")
