"""GPU: split strips (round 4, sw_systolic2.inc).  From some strip on, the strip's scout -- its producer the scout's loop down to the split
and the filler's below it -- writes the strip's lower 16-row blocks and the filler only the upper ones.  Forced here (options split_blk /
split_from) at every kind of split point on shapes the oracle fills in full; the library's own choice runs in the 16384^2 tests."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def strips_every_126(engine):
    """(split strips belong to the scouts' regime of 126-column strips; an int64 H of 167 strips would otherwise run as overlapping strips)"""
    engine.set_option("s2w", 126)
    yield
    engine.set_option("s2w", 0)


def _roles_per_xcd(engine):
    if not engine.get_option("xcd_round_robin"):
        pytest.skip("workgroups are not dealt round-robin to 8 XCDs here")


@pytest.mark.parametrize("strips,rows,blk,frm", [(40, 333, 3, 1), (40, 333, 20, 1), (131, 272, 9, 100), (131, 272, 1, 1), (64, 1000, 40, 30),
                                                 (17, 200, 12, 1), (129, 144, 5, 120), (40, 330, 20, 39), (167, 160, 4, 1)])
def test_forced_splits_match_the_oracle(engine, oracle, strips, rows, blk, frm):
    import torch
    _roles_per_xcd(engine)
    a, b = oracle.generate(126 * strips - (strips % 3), rows, strips + rows)
    H, P, mp = oracle.fill(a, b)
    engine.set_option("split_blk", blk)
    engine.set_option("split_from", frm)
    try:
        for kw in ({}, {"p_dtype": torch.int8}, {"h_dtype": torch.int64}, {"want_h": False}, {"want_h": False, "want_p": False}):
            if (126 * strips - (strips % 3)) % 2 and kw:
                continue                                  # (an odd number of columns: int32 H + P only)
            out = engine.fill(a, b, **kw)
            assert engine.get_option("last_xcd_mode") == 1 and engine.get_option("last_split_from") == frm, kw
            r = out.result()
            assert (r["max_pos"], r["max_score"]) == (mp, int(H.flat[mp])), kw
            if out.H is not None:
                assert np.array_equal(out.H.cpu().numpy().astype(np.int32), H), kw
            if out.P is not None:
                assert np.array_equal(out.P.cpu().numpy().astype(np.int32), P), kw
    finally:
        engine.set_option("split_blk", 0)
        engine.set_option("split_from", 0)


def test_library_chosen_split_at_8192(engine, oracle, swamd):
    """rows >= 4096: the library splits the last strips by itself"""
    _roles_per_xcd(engine)
    a, b = swamd.generate(8192, 8192, 3)
    out = engine.fill(a, b)
    assert engine.get_option("last_split_from") > 0
    st = oracle.fill_streaming(a, b)
    r = out.result()
    assert (r["max_pos"], r["max_score"]) == (st["max_pos"], st["max_score"])
    assert np.array_equal(engine.row_checksums(out.H), st["csH"]) and np.array_equal(engine.row_checksums(out.P), st["csP"])
    engine.set_option("debug_flags", 1048576)            # and without
    try:
        out2 = engine.fill(a, b)
        assert engine.get_option("last_split_from") == 0
        assert np.array_equal(engine.row_checksums(out2.H), st["csH"]) and np.array_equal(engine.row_checksums(out2.P), st["csP"])
    finally:
        engine.set_option("debug_flags", 0)
