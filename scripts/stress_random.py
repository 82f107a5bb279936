# Random shapes, grids and output modes on whichever kernel the library picks, every fill (and a third of the tracebacks) against
# the oracle; then random batches on the one-pair-per-wave kernel.  Run on the GPU box (SW_STRESS_SECONDS, default 170 + 60).
# round 2: 82 149 fills in 7 minutes, no mismatch.  round 3 adds rows that are not multiples of 16, scouts, tracebacks, batches; round 4 wide
# matrices (overlapping strips, column tiles), forced overlapping strips, the packed batch kernels (score-only and int8 P).
import importlib, sys, os, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
sw = importlib.import_module("smith-waterman_amd")
import oracle_lib
orc = oracle_lib.Oracle()
eng = sw.Engine(0)
rng = np.random.default_rng(int(os.environ.get("SW_STRESS_SEED", "2025")))
budget = float(os.environ.get('SW_STRESS_SECONDS', '170'))
t0 = time.time(); tp = t0; n = n2 = nsc = nx = ntb = bad = 0
modes = ["hp", "p8", "p8_only", "p32_only", "h_only", "score_only", "h64", "h64_p8"]
while time.time() - t0 < budget:
    cols = int(rng.integers(1, 1600)) * 2 if rng.random() < 0.8 else int(rng.integers(1, 3000))
    rows = int(rng.integers(1, 60)) * 16 if rng.random() < 0.5 else int(rng.integers(1, 900))
    if rng.random() < 0.15: cols, rows = int(rng.integers(2000, 21500)), int(rng.integers(1, 200))   # up to 171 strips: per-XCD roles, two-strip scouts
    if rng.random() < 0.06: cols, rows = int(rng.integers(18000, 40000)), int(rng.integers(1, 150))  # beyond the scouts: overlapping strips (even widths) / column tiles (odd)
    if rng.random() < 0.10: eng.set_option("s2w", 110)                                               # overlapping strips forced on any shape
    mode = modes[int(rng.integers(0, len(modes)))]
    a, b = orc.generate(cols, rows, int(rng.integers(1, 1 << 30)))
    if rng.random() < 0.2: eng.set_option("max_blocks", int(rng.integers(1, 20)))
    want_h = mode in ("hp", "p8", "h_only", "h64", "h64_p8"); want_p = mode in ("hp", "p8", "p8_only", "p32_only", "h64", "h64_p8")
    out = eng.fill(a, b, h_dtype=torch.int64 if mode.startswith("h64") else None, p_dtype=torch.int8 if "p8" in mode else None, want_h=want_h, want_p=want_p)
    eng.set_option("max_blocks", 0); eng.set_option("s2w", 0)
    two = eng.get_option("last_strips2") > 0
    n2 += two; nsc += eng.get_option("last_scouts") > 0; nx += eng.get_option("last_xcd_mode") > 0
    H, P, mp = orc.fill(a, b)
    r = out.result()
    ok = r["max_pos"] == mp and r["max_score"] == int(H.flat[mp])
    if want_h: ok = ok and np.array_equal(out.H.cpu().numpy().astype(np.int64), H.astype(np.int64))
    if want_p: ok = ok and np.array_equal(out.P.cpu().numpy().astype(np.int32), P)
    if ok and want_p and rng.random() < 0.33:
        path = eng.traceback(out, mp)
        ok = np.array_equal(path, orc.backtrack(P, mp)) and np.array_equal(out.P.cpu().numpy().astype(np.int32), P)
        ntb += 1
    n += 1
    if not ok:
        bad += 1
        print("MISMATCH", cols, rows, mode, "two-col" if two else "one-col", flush=True)
        if bad > 5: break
    if time.time() - tp > 30: tp = time.time(); print(f"... {n} fills, {bad} mismatches", flush=True)
print(f"{n} random fills in random output modes ({n2} on the two-column kernel, {nsc} with scouts, {nx} with roles dealt per XCD, {ntb} traced back), {bad} mismatches", flush=True)
# ---- batches
t0 = time.time(); nb = npairs_total = badb = 0
letters = np.frombuffer(b"ACGTNRYK", np.uint8)
while time.time() - t0 < float(os.environ.get('SW_STRESS_BATCH_SECONDS', '60')) and badb <= 5:
    cols, rows, npairs = int(rng.integers(1, 2300)), int(rng.integers(1, 400)), int(rng.integers(1, 40))
    nl = int(rng.integers(1, 9))
    A = letters[rng.integers(0, nl, (npairs, cols))]; B = letters[rng.integers(0, nl, (npairs, rows))]
    sc = [(3, -3, -2), (5, -3, -4), (2, 1, -1), (1, -1, 0)][int(rng.integers(0, 4))]
    mode = int(rng.integers(0, 4))   # 0 score only, 1 H + P int32, 2 int8 P only, 3 H + int8 P, traceback for 2
    res, H, P = eng.batch(A, B, scores=sc, store=mode > 0, p_dtype=torch.int8 if mode >= 2 else None, store_h=mode in (1, 3), traceback=mode == 2)
    assert eng.get_option("last_batch_kernel") in (1, 2)
    npk = npk + (eng.get_option("last_batch_kernel") == 2) if "npk" in dir() else int(eng.get_option("last_batch_kernel") == 2)
    res = res.cpu().numpy()
    for k in range(npairs):
        h, p, mp = orc.fill(A[k], B[k], sc)
        ok = res[k, 0] == mp and res[k, 1] == int(h.flat[mp])
        if H is not None: ok = ok and np.array_equal(H[k].cpu().numpy(), h)
        if mode == 2:
            path = orc.backtrack(p, mp)
            ok = ok and res[k, 2] == len(path)
        if P is not None: ok = ok and np.array_equal(P[k].cpu().numpy().astype(np.int32), p)
        if not ok:
            badb += 1; print("BATCH MISMATCH", cols, rows, npairs, nl, sc, mode, k, flush=True)
    nb += 1; npairs_total += npairs
    if time.time() - tp > 30: tp = time.time(); print(f"... {nb} batches, {badb} mismatches", flush=True)
print(f"{nb} random batches ({npairs_total} pairs; {npk} batches on the packed two-pairs-per-wave kernels), {badb} mismatches", flush=True)
