"""GPU: tile decomposition (multi-GPU building block), batched pairs, and the band pipeline on one GPU."""
import importlib
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_tile_decomposition_gpu(engine, oracle):
    """2 row bands x 3 column chunks through sw_fill_tile_device, written into ONE matrix, equal the serial fill."""
    import torch
    cols, rows = 700, 500
    a, b = oracle.generate(cols, rows, 21)
    H, P, mp = oracle.fill(a, b)
    d_a, _ = engine.to_device(a)
    d_b, _ = engine.to_device(b)
    dH = torch.zeros((rows + 1, cols + 1), dtype=torch.int32, device="cuda")
    dP = torch.zeros_like(dH)
    res = torch.zeros(3, dtype=torch.int64, device="cuda")
    bands = [(0, 256), (256, 500)]
    chunks = [(0, 252), (252, 504), (504, 700)]
    best = (0, 0)
    for (i0, i1) in bands:
        left = None
        for k, (j0, j1) in enumerate(chunks):
            top = dH[i0, j0:j1 + 1].clone() if i0 > 0 else None
            right = torch.zeros(i1 - i0 + 1, dtype=torch.int32, device="cuda")
            engine.fill_tile(dH, dP, i0, j0, i1 - i0, j1 - j0, d_a, d_b, res, top=top, left=left, right=right)
            engine.synchronize()
            r = res.cpu().tolist()
            assert np.array_equal(right.cpu().numpy(), H[i0:i1 + 1, j1]), "right edge column"
            if r[1] > 0:
                gpos = r[0] + i0 * (cols + 1) + j0
                if r[1] > best[0] or (r[1] == best[0] and gpos < best[1]):
                    best = (r[1], gpos)
            left = right
    assert np.array_equal(dH.cpu().numpy(), H)
    assert np.array_equal(dP.cpu().numpy(), P)
    assert best == (int(H.flat[mp]), mp)


def test_batch_pairs(engine, oracle):
    rng = np.random.default_rng(3)
    npairs, cols, rows = 23, 200, 150
    A = (rng.integers(0, 4, (npairs, cols)) + 65).astype(np.uint8)
    B = (rng.integers(0, 4, (npairs, rows)) + 65).astype(np.uint8)
    res, H, P = engine.batch(A, B, store=True)
    res = res.cpu().numpy()
    for k in range(npairs):
        h, p, mp = oracle.fill(A[k], B[k])
        assert np.array_equal(H[k].cpu().numpy(), h) and np.array_equal(P[k].cpu().numpy(), p), f"pair {k}"
        assert res[k, 0] == mp and res[k, 1] == int(h.flat[mp])
    res2, _, _ = engine.batch(A, B, store=False)   # score-only
    assert np.array_equal(res2.cpu().numpy()[:, 1], res[:, 1])


def test_batch_1024_pairs_config5_shape(engine, oracle, swamd):
    """BASELINE config 5 shape (1024 x 1024 pairs), a small batch: pair k uses seed 1+k."""
    npairs = 6
    A = np.stack([swamd.generate(1024, 1024, 1 + k)[0] for k in range(npairs)])
    B = np.stack([swamd.generate(1024, 1024, 1 + k)[1] for k in range(npairs)])
    res, H, P = engine.batch(A, B, store=True)
    res = res.cpu().numpy()
    h = oracle_hashes = None
    from oracle_lib import golden_hashes
    g = golden_hashes()["rand_1024x1024_s1"]
    assert res[0, 0] == g["maxPos"] and res[0, 1] == g["maxScore"]
    assert f"{oracle.fnv(H[0].cpu().numpy()):016x}" == g["fnvH"] and f"{oracle.fnv(P[0].cpu().numpy()):016x}" == g["fnvP0"]
    for k in (1, npairs - 1):
        hh, pp, mp = oracle.fill(A[k], B[k])
        assert np.array_equal(H[k].cpu().numpy(), hh) and np.array_equal(P[k].cpu().numpy(), pp) and res[k, 0] == mp


def test_batch_at_scale_shape_small_pairs(engine, oracle):
    """5000 small pairs (ragged: 130 columns on 4-column lanes) through the one-pair-per-wave kernel, and once more through the
    single-pair machinery (debug bit 16: the launch shape config 5 used before round 3 -- two strips per workgroup, several
    passes of the resident grid, more than one 4096-pair chunk): EVERY pair's H, P and arg-max against the oracle."""
    rng = np.random.default_rng(11)
    npairs, cols, rows = 5000, 130, 70
    A = (rng.integers(0, 4, (npairs, cols)) + 65).astype(np.uint8)
    B = (rng.integers(0, 4, (npairs, rows)) + 65).astype(np.uint8)
    res, H, P = engine.batch(A, B, store=True)
    assert engine.get_option("last_batch_kernel") == 1
    engine.set_option("debug_flags", 65536)
    try:
        res_o, H_o, P_o = engine.batch(A, B, store=True)
        assert engine.get_option("last_batch_kernel") == 0
        assert engine.get_option("last_grid") == engine.get_option("num_cus")   # capped grid: many passes per workgroup
    finally:
        engine.set_option("debug_flags", 0)
    assert bool((res_o == res).all()) and bool((H_o == H).all()) and bool((P_o == P).all())
    del res_o, H_o, P_o
    res, H, P = res.cpu().numpy(), H.cpu().numpy(), P.cpu().numpy()
    assert (res[:, 2] == 0).all()
    for k in range(npairs):
        h, p, mp = oracle.fill(A[k], B[k])
        assert np.array_equal(H[k], h) and np.array_equal(P[k], p), f"pair {k}"
        assert res[k, 0] == mp and res[k, 1] == int(h.flat[mp]), f"pair {k} arg-max"
    res2, _, _ = engine.batch(A, B, store=False)   # score-only: same scores, same exact arg-max
    assert np.array_equal(res2.cpu().numpy(), res)


def test_batch_300_pairs_of_1024_vs_oracle(engine, oracle, swamd):
    """BASELINE config 5's pair shape (1024 x 1024, pair k seeded 1+k), 300 pairs = 5100 strips (20 passes of the grid):
    H, P, arg-max of every pair against the oracle."""
    npairs = 300
    gen = [swamd.generate(1024, 1024, 1 + k) for k in range(npairs)]
    A, B = np.stack([g[0] for g in gen]), np.stack([g[1] for g in gen])
    res, H, P = engine.batch(A, B, store=True)
    res = res.cpu().numpy()
    for k in range(npairs):
        h, p, mp = oracle.fill(A[k], B[k])
        assert np.array_equal(H[k].cpu().numpy(), h) and np.array_equal(P[k].cpu().numpy(), p), f"pair {k}"
        assert res[k, 0] == mp and res[k, 1] == int(h.flat[mp]), f"pair {k} arg-max"
    res2, _, _ = engine.batch(A, B, store=False)
    assert np.array_equal(res2.cpu().numpy(), res)


def test_batch_compact_p_and_traceback(engine, oracle):
    """BASELINE config 5's deliverable per pair -- score, maxPos, path -- with int8 P and no H (5x less memory):
    sw_batch_device_ex + sw_batch_traceback_device against the oracle's fill + backtrack for every pair."""
    import torch
    rng = np.random.default_rng(5)
    npairs, cols, rows = 700, 200, 150
    A = (rng.integers(0, 4, (npairs, cols)) + 65).astype(np.uint8)
    B = (rng.integers(0, 4, (npairs, rows)) + 65).astype(np.uint8)
    A[3], B[3] = 65, 67          # a pair without any match: H == 0, maxPos 0, empty path
    res, H, P, paths = engine.batch(A, B, store=True, p_dtype=torch.int8, store_h=False, traceback=True, want_paths=True)
    assert H is None and P.dtype == torch.int8
    res, P, paths = res.cpu().numpy(), P.cpu().numpy(), paths.cpu().numpy()
    for k in range(npairs):
        h, p, mp = oracle.fill(A[k], B[k])
        opath = oracle.backtrack(p, mp)   # p is negated along the path now
        assert res[k, 0] == mp and res[k, 1] == int(h.flat[mp]) and res[k, 2] == len(opath), f"pair {k}"
        assert np.array_equal(paths[k, :len(opath)], opath) and np.array_equal(P[k].astype(np.int32), p), f"pair {k} path"


def test_band_pipeline_single_gpu(engine, oracle):
    """BandPipeline with the GPU tile engine, world_size 1 (several chunks): exercises GpuTiles end to end."""
    import torch.distributed as dist
    multi = importlib.import_module("smith-waterman_amd.multi")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29611")
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        a, b = oracle.generate(900, 400, 13)
        H, P, mp = oracle.fill(a, b)
        pipe = multi.BandPipeline(dist, 0, 1, a, b, nchunks=4, make_tiles=lambda *x: multi.GpuTiles(engine, *x))
        score, pos = pipe.fill()
        assert (score, pos) == (int(H.flat[mp]), mp)
        n = pipe.traceback(pos)
        path = oracle.backtrack(P, mp)
        dH, dP = pipe.tiles.matrices()
        assert n == len(path) and np.array_equal(dH, H) and np.array_equal(dP, P)
    finally:
        dist.destroy_process_group()


def test_config5_at_scale_100000_pairs_sampled(engine, oracle, swamd):
    """BASELINE config 5 at FULL scale: 100 000 pairs of 1024 x 1024 (pair k seeded 1+k) in one call, int8 P for every
    pair (105 GB resident), per-pair traceback; 1000 sampled pairs are checked against the oracle: score, maxPos, path
    length, and the whole (negated) P matrix for 50 of them."""
    import torch
    free, _ = torch.cuda.mem_get_info()
    if free < (125 << 30):
        pytest.skip(f"needs 125 GB of free HBM, {free >> 30} GB free")
    npairs = 100000
    A = np.empty((npairs, 1024), np.uint8)
    B = np.empty((npairs, 1024), np.uint8)
    for k in range(npairs):
        A[k], B[k] = swamd.generate(1024, 1024, 1 + k)
    res, H, P = engine.batch(A, B, store=True, p_dtype=torch.int8, store_h=False, traceback=True)
    res = res.cpu().numpy()
    rng = np.random.default_rng(7)
    sample = np.unique(np.concatenate([[0, 1, npairs - 1, 4095, 4096, 4097], rng.integers(0, npairs, 1000)]))
    for n, k in enumerate(sample):
        h, p, mp = oracle.fill(A[k], B[k])
        path = oracle.backtrack(p, mp)
        assert res[k, 0] == mp and res[k, 1] == int(h.flat[mp]) and res[k, 2] == len(path), f"pair {k}"
        if n % 20 == 0:
            assert np.array_equal(P[k].cpu().numpy().astype(np.int32), p), f"pair {k} P"
    del P
    torch.cuda.empty_cache()
