"""GPU: overlapping strips of the two-column kernel (round 4): strips every 110 columns instead of 126, so that every 64-byte line of a
matrix row lies wholly inside one strip and is stored by ONE instruction (whole-line stores).  Forced here with option "s2w" on shapes
the oracle fills in full, in every output format; the library's own choice runs in the at-size tests."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture()
def overlap(engine):
    engine.set_option("s2w", 110)
    yield engine
    engine.set_option("s2w", 0)


@pytest.mark.parametrize("cols,rows,seed", [(126, 40, 1), (236, 33, 2), (5040, 333, 3), (4400, 272, 4), (16500, 144, 5), (1000, 1000, 6), (346, 17, 7), (14410, 64, 8),
                                            (2, 16, 9), (110, 16, 10), (112, 100, 11), (20000, 50, 12),
                                            (5041, 333, 13), (4401, 100, 14), (16501, 144, 15), (347, 17, 16), (237, 33, 17), (1001, 1000, 18), (127, 40, 19), (20001, 50, 20)])
def test_overlapping_strips_match_the_oracle(overlap, oracle, cols, rows, seed):
    import torch
    engine = overlap
    a, b = oracle.generate(cols, rows, seed)
    H, P, mp = oracle.fill(a, b)
    for kw in ({}, {"p_dtype": torch.int8}, {"h_dtype": torch.int64}, {"want_h": False}, {"want_h": False, "want_p": False}, {"want_p": False}):
        out = engine.fill(a, b, **kw)
        if engine.get_option("last_tiles") == 1 and (cols % 2 == 0 or kw == {}):
            # (an odd width runs on the two-column kernel only with int32 H + int32 P -- and then with whole-line stores, every row shifted)
            assert engine.get_option("last_strips2") == (1 if cols <= 126 else -(-(cols - 126) // 110) + 1), kw
        r = out.result()
        assert (r["max_pos"], r["max_score"]) == (mp, int(H.flat[mp])), kw
        if out.H is not None:
            assert np.array_equal(out.H.cpu().numpy().astype(np.int32), H), kw
        if out.P is not None:
            assert np.array_equal(out.P.cpu().numpy().astype(np.int32), P), kw
        if out.P is not None and out.H is not None:
            path = engine.traceback(out, mp)
            assert np.array_equal(path, oracle.backtrack(P.copy(), mp)), kw


def test_overlapping_strips_ties_and_repeats(overlap, oracle):
    engine = overlap
    a = np.tile(np.frombuffer(b"ACGT", np.uint8), 1500)
    b = np.tile(np.frombuffer(b"ACGT", np.uint8), 80)
    for _ in range(3):
        out = engine.fill(a, b)
        H, P, mp = oracle.fill(a, b)
        assert out.result()["max_pos"] == mp and np.array_equal(out.H.cpu().numpy(), H) and np.array_equal(out.P.cpu().numpy(), P)


def test_overlapping_strips_tiles_and_splits(overlap, oracle, swamd):
    """column tiles (30000 columns) and forced split strips on the overlapping geometry"""
    engine = overlap
    a, b = swamd.generate(30000, 600, 2)
    H, P, mp = oracle.fill(a, b)
    out = engine.fill(a, b)
    assert engine.get_option("last_tiles") >= 2
    assert out.result()["max_pos"] == mp and np.array_equal(out.H.cpu().numpy(), H) and np.array_equal(out.P.cpu().numpy(), P)
    a, b = oracle.generate(110 * 60, 400, 5)
    H, P, mp = oracle.fill(a, b)
    engine.set_option("split_blk", 7); engine.set_option("split_from", 1)
    try:
        out = engine.fill(a, b)
        assert engine.get_option("last_split_from") == 1
        assert out.result()["max_pos"] == mp and np.array_equal(out.H.cpu().numpy(), H) and np.array_equal(out.P.cpu().numpy(), P)
    finally:
        engine.set_option("split_blk", 0); engine.set_option("split_from", 0)


def test_overlapping_strips_16384_streaming_checksums(overlap, oracle, swamd):
    engine = overlap
    n = 16384
    a, b = swamd.generate(n, n, 1)
    out = engine.fill(a, b)
    st = oracle.fill_streaming(a, b)
    r = out.result()
    assert (r["max_pos"], r["max_score"]) == (st["max_pos"], st["max_score"])
    assert np.array_equal(engine.row_checksums(out.H), st["csH"]) and np.array_equal(engine.row_checksums(out.P), st["csP"])


def test_overlapping_strips_are_the_librarys_choice_beyond_the_scouts(engine, oracle):
    """more than 170 strips of 126 columns (no room for scouts): one launch of 110-column strips that stream whole lines, int32 and int64 H alike
    (200 strips for 22000 columns); an int64 H already where its 110-column strips no longer fit beside scouts (20000 columns: 182 strips);
    odd widths and other formats keep the 126-column geometry.  The arg-max of a strip is looked up among the cells that strip stored
    (rows that end inside a block, a maximum in the overlap columns)"""
    import torch
    for cols, rows, seed in ((22000, 50, 12), (22000, 333, 5), (21560, 17, 3)):
        a, b = oracle.generate(cols, rows, seed)
        H, P, mp = oracle.fill(a, b)
        for kw in ({"h_dtype": torch.int64}, {}):
            out = engine.fill(a, b, **kw)
            assert engine.get_option("last_strips2") == -(-(cols - 126) // 110) + 1 and engine.get_option("last_tiles") == 1, kw
            r = out.result()
            assert (r["max_pos"], r["max_score"]) == (mp, int(H.flat[mp])), kw
            assert np.array_equal(out.H.cpu().numpy().astype(np.int64), H.astype(np.int64)) and np.array_equal(out.P.cpu().numpy(), P), kw
    a, b = oracle.generate(20000, 40, 2)
    H, P, mp = oracle.fill(a, b)
    out = engine.fill(a, b, h_dtype=torch.int64)
    assert engine.get_option("last_strips2") == 182 and engine.get_option("last_scouts") == 0 and out.result()["max_pos"] == mp
    assert np.array_equal(out.H.cpu().numpy(), H.astype(np.int64)) and np.array_equal(out.P.cpu().numpy(), P)
    out = engine.fill(a, b)                                  # int32 H: 159 strips of 126 columns behind scouts
    assert engine.get_option("last_strips2") == 159 and out.result()["max_pos"] == mp and np.array_equal(out.P.cpu().numpy(), P)
    a, b = oracle.generate(22001, 40, 2)                     # an odd width: every row stores shifted pairs
    H, P, mp = oracle.fill(a, b)
    out = engine.fill(a, b)
    assert engine.get_option("last_tiles") == 1 and engine.get_option("last_strips2") == 200 and out.result()["max_pos"] == mp
    assert np.array_equal(out.H.cpu().numpy(), H) and np.array_equal(out.P.cpu().numpy(), P)
    out = engine.fill(a, b, p_dtype=torch.int8)             # ... an odd width with int8 P: the one-column kernel
    assert engine.get_option("last_strips2") == 0 and out.result()["max_pos"] == mp and np.array_equal(out.P.cpu().numpy().astype(np.int32), P)
    a, b = oracle.generate(22000, 40, 2)                     # int32 H + int8 P: whole lines of H, the P bytes as ever
    H, P, mp = oracle.fill(a, b)
    out = engine.fill(a, b, p_dtype=torch.int8)
    assert engine.get_option("last_strips2") == 200 and out.result()["max_pos"] == mp
    assert np.array_equal(out.H.cpu().numpy(), H) and np.array_equal(out.P.cpu().numpy().astype(np.int32), P)
    out = engine.fill(a, b, p_dtype=torch.int8, want_h=False)   # P only: 126-column strips
    assert engine.get_option("last_strips2") == 175 and out.result()["max_pos"] == mp and np.array_equal(out.P.cpu().numpy().astype(np.int32), P)


def test_overlapping_strips_maximum_in_every_overlap_column(overlap, oracle):
    """a planted perfect match that ends in column c for every c around a strip boundary: the cell is computed by two strips and reported by the
    one that stores it"""
    rng = np.random.default_rng(5)
    cols, rows = 700, 96
    for c_end in range(215, 240):
        a = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, cols)].copy()
        b = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, rows)].copy()
        L = 40
        r_end = int(rng.integers(L, rows + 1))
        b[r_end - L:r_end] = a[c_end - L:c_end]
        H, P, mp = oracle.fill(a, b)
        for kw in ({}, {"h_dtype": __import__("torch").int64}):
            out = overlap.fill(a, b, **kw)
            r = out.result()
            assert (r["max_pos"], r["max_score"]) == (mp, int(H.flat[mp])), (c_end, kw)
