import importlib, sys, ctypes, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from oracle_lib import golden
sw = importlib.import_module("smith-waterman_amd")
names = ["rand_256x256_s1", "rand_300x200_s1", "rand_129x64_s11"]
dims = {"rand_256x256_s1": (256, 256), "rand_300x200_s1": (300, 200), "rand_129x64_s11": (129, 64)}
gs = {nm: golden(nm) for nm in names}
eng = sw.Engine(0)
k = 0; shown = 0
for it in range(int(sys.argv[1])):
    for nm in names:
        g = gs[nm]; k += 1
        out = eng.fill(g["a"], g["b"])
        dH = out.H.cpu().numpy()
        if not np.array_equal(dH, g["H"]) and shown < 4:
            shown += 1
            ep = ((k - 1) % 255) + 1
            cols, rows = dims[nm]; S = (cols + 62) // 63; e4s = ((rows + S + 160 + 3) // 4) * 4
            cap = eng.get_option("debug_edge4_cap"); ptr = eng.get_option("debug_edge4_ptr")
            buf = np.zeros(cap, np.uint32)
            sw._check(sw.lib().sw_memcpy_d2h(eng._h, buf.ctypes.data, ptr, cap * 4))
            bad = np.argwhere(dH != g["H"]); r, c = bad[0]
            print(f"fill {k} epoch {ep} {nm}: first bad ({r},{c}); e4stride {e4s}, cap {cap}")
            tags = buf >> 24
            for s in range(S):
                row = buf[s * e4s:(s + 1) * e4s]
                cur = row[(row >> 24) == ep]
                pay = (cur & 0xffffff).astype(np.int64) - 0x10000
                weird = np.argwhere(((row >> 24) == ep) & (((row & 0xffffff).astype(np.int64) - 0x10000 > 20000) | ((row & 0xffffff).astype(np.int64) < 0x8000))).ravel()
                print(f"  strip {s}: {len(cur)} words with tag {ep}; payload range {pay.min() if len(pay) else 0}..{pay.max() if len(pay) else 0}; weird idx {weird[:8].tolist()} vals {[hex(int(x)) for x in row[weird[:4]]]}")
            other = np.bincount(tags, minlength=256)
            print("  tag histogram (nonzero):", {int(t): int(n) for t, n in enumerate(other) if n and t != 0}, "zeros:", int(other[0]))
