"""GPU: column tiles of the two-column kernel (round 4).  A matrix too wide for scout workgroups beside one filler per strip (more than
~21 000 columns) is filled by several launches, one per column tile, each with scouts; a tile's left halo is the previous tile's last
column of H, the arg-max accumulates across the launches.  Everything must equal the oracle cell for cell -- and the untiled launch."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def strips_every_126(engine):
    """(since the overlapping strips with streaming whole-line stores the library fills wide matrices of even width in ONE launch; the tiles
    remain for the 126-column geometry -- odd widths, or option s2w = 126 as here)"""
    engine.set_option("s2w", 126)
    yield
    engine.set_option("s2w", 0)


def _fill(engine, a, b, **kw):
    out = engine.fill(a, b, **kw)
    return out, out.result(), int(engine.get_option("last_tiles"))


@pytest.mark.parametrize("cols,rows,seed", [(30000, 2000, 1), (21500, 1000, 2), (30001, 777, 3), (35000, 300, 4)])
def test_tiled_fill_matches_oracle(engine, oracle, swamd, cols, rows, seed):
    a, b = swamd.generate(cols, rows, seed)
    out, r, nt = _fill(engine, a, b)
    assert nt >= 2, "expected the column-tile path"
    H, P, mp = oracle.fill(a, b)
    assert np.array_equal(out.H.cpu().numpy(), H) and np.array_equal(out.P.cpu().numpy(), P)
    assert (r["max_pos"], r["max_score"]) == (mp, int(H.flat[mp]))
    path = engine.traceback(out, mp)
    assert np.array_equal(path, oracle.backtrack(P, mp))
    # the untiled launch (debug bit 19) gives the same
    engine.set_option("debug_flags", 524288)
    try:
        out2, r2, nt2 = _fill(engine, a, b)
    finally:
        engine.set_option("debug_flags", 0)
    assert nt2 == 1 and r2["max_pos"] == mp and np.array_equal(out2.H.cpu().numpy(), H)


def test_tiles_ties_and_maximum_in_a_later_tile(engine, oracle):
    """periodic sequences: the maximum is attained in many cells of several tiles (the lowest linear index wins); then a pair whose only
    good alignment lies in the last tile"""
    cols, rows = 26000, 600
    a = np.tile(np.frombuffer(b"ACGT", np.uint8), cols // 4)
    b = np.tile(np.frombuffer(b"ACGT", np.uint8), rows // 4)
    out, r, nt = _fill(engine, a, b)
    H, P, mp = oracle.fill(a, b)
    assert nt >= 2 and r["max_pos"] == mp and np.array_equal(out.H.cpu().numpy(), H) and np.array_equal(out.P.cpu().numpy(), P)
    rng = np.random.default_rng(5)
    a = np.frombuffer(b"AC", np.uint8)[rng.integers(0, 2, cols)].copy()
    b = np.frombuffer(b"GT", np.uint8)[rng.integers(0, 2, rows)].copy()
    a[-500:] = np.resize(b, 500)                      # the only matches: the last 500 columns
    out, r, nt = _fill(engine, a, b)
    H, P, mp = oracle.fill(a, b)
    assert nt >= 2 and mp % (cols + 1) > cols - 501 and r["max_pos"] == mp and r["max_score"] == int(H.flat[mp])
    assert np.array_equal(out.H.cpu().numpy(), H) and np.array_equal(out.P.cpu().numpy(), P)


def test_tiles_with_an_alphabet_of_more_than_7_letters_fall_back_as_a_whole(engine, oracle):
    """every tile's prologue scans the WHOLE a: all tiles decide alike, and the fall-back kernel behind them fills the whole matrix"""
    rng = np.random.default_rng(9)
    alpha = np.frombuffer(b"ACGTNRYKM", np.uint8)
    cols, rows = 24000, 400
    a = alpha[rng.integers(0, 4, cols)].copy()
    a[-3000:] = alpha[rng.integers(0, 9, 3000)]       # the extra letters only occur in the last tile
    b = alpha[rng.integers(0, 4, rows)].copy()
    out = engine.fill(a, b)
    H, P, mp = oracle.fill(a, b)
    assert out.result()["max_pos"] == mp and np.array_equal(out.H.cpu().numpy(), H) and np.array_equal(out.P.cpu().numpy(), P)


def test_tiled_32768_square_streaming_checksums(engine, oracle, swamd):
    """the size the tiles were made for: 32768 x 32768 (two tiles of 16384 columns) against the streaming oracle"""
    import torch
    if torch.cuda.mem_get_info()[0] < (20 << 30):
        pytest.skip("needs 20 GB of free HBM")
    n = 32768
    a, b = swamd.generate(n, n, 1)
    out, r, nt = _fill(engine, a, b)
    assert nt == 2
    st = oracle.fill_streaming(a, b)
    assert (r["max_pos"], r["max_score"]) == (st["max_pos"], st["max_score"])
    assert np.array_equal(engine.row_checksums(out.H), st["csH"]) and np.array_equal(engine.row_checksums(out.P), st["csP"])
