# Random shapes, grids and output modes on whichever kernel the library picks, every fill against the oracle (run on the GPU box;
# round 2: 82 149 fills in 7 minutes -- eight output modes incl. int64 H --, 60 976 of them on the two-column kernel, no mismatch).
import importlib, sys, os, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
sw = importlib.import_module("smith-waterman_amd")
import oracle_lib
orc = oracle_lib.Oracle()
eng = sw.Engine(0)
rng = np.random.default_rng(2024)
t0 = time.time(); n = 0; n2 = 0; bad = 0
modes = ["hp", "p8", "p8_only", "p32_only", "h_only", "score_only", "h64", "h64_p8"]
while time.time() - t0 < float(os.environ.get('SW_STRESS_SECONDS', '170')):
    cols = int(rng.integers(1, 1600)) * 2 if rng.random() < 0.8 else int(rng.integers(1, 3000))
    rows = int(rng.integers(1, 60)) * 16 if rng.random() < 0.8 else int(rng.integers(1, 900))
    mode = modes[int(rng.integers(0, len(modes)))]
    a, b = orc.generate(cols, rows, int(rng.integers(1, 1 << 30)))
    if rng.random() < 0.2: eng.set_option("max_blocks", int(rng.integers(1, 20)))
    want_h = mode in ("hp", "p8", "h_only", "h64", "h64_p8"); want_p = mode in ("hp", "p8", "p8_only", "p32_only", "h64", "h64_p8")
    out = eng.fill(a, b, h_dtype=torch.int64 if mode.startswith("h64") else None, p_dtype=torch.int8 if "p8" in mode else None, want_h=want_h, want_p=want_p)
    eng.set_option("max_blocks", 0)
    two = eng.get_option("last_strips2") > 0
    n2 += two
    H, P, mp = orc.fill(a, b)
    r = out.result()
    ok = r["max_pos"] == mp and r["max_score"] == int(H.flat[mp])
    if want_h: ok = ok and np.array_equal(out.H.cpu().numpy().astype(np.int64), H.astype(np.int64))
    if want_p: ok = ok and np.array_equal(out.P.cpu().numpy().astype(np.int32), P)
    n += 1
    if not ok:
        bad += 1
        print("MISMATCH", cols, rows, mode, "two-col" if two else "one-col")
        if bad > 5: break
print(f"{n} random fills in random output modes ({n2} on the two-column kernel), {bad} mismatches")
