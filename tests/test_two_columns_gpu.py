"""GPU: the two-columns-per-lane fill kernel (sw_systolic2.inc) -- selected by the library for whole-matrix fills of one pair
with up to 7 letters -- against the oracle, bit-exact, and against the one-column kernel (debug bit 14)."""
import numpy as np
import pytest

from test_fill_gpu import check_against_oracle

pytestmark = pytest.mark.gpu

# strips are 126 columns wide: single strip, one column over, odd column counts (the last strip then has a lane that owns only
# its A column), many strips; rows: one block, several, not a multiple of the ring length
SHAPES = [(1, 16), (2, 16), (3, 32), (125, 48), (126, 16), (127, 160), (251, 64), (252, 64), (253, 80), (1000, 704), (1007, 304),
          (4000, 1296), (5001, 528)]


@pytest.mark.parametrize("cols,rows", SHAPES)
def test_two_column_kernel_matches_oracle(engine, oracle, cols, rows):
    a, b = oracle.generate(cols, rows, 77)
    check_against_oracle(engine, oracle, a, b)
    assert engine.get_option("last_strips2") == (cols + 125) // 126, "the two-column kernel was expected to run"


@pytest.mark.parametrize("cols,rows", [(500, 250), (126, 1), (127, 15), (1000, 17), (2000, 2001), (333, 63), (4001, 1295), (150000 // 8, 2000)])
def test_rows_not_a_multiple_of_16(engine, oracle, cols, rows):
    """a short last block: its rows below the matrix are computed and their stores dropped (the reference's own sweep,
    run-v0.sh:30-38, is 2000 rows x {2 .. 150000} columns)"""
    a, b = oracle.generate(cols, rows, 78)
    check_against_oracle(engine, oracle, a, b)
    assert engine.get_option("last_strips2") == (cols + 125) // 126


@pytest.mark.parametrize("mode", ["p8", "p8_only", "score_only", "h64"])
def test_rows_not_a_multiple_of_16_other_formats(engine, oracle, mode):
    import torch
    a, b = oracle.generate(1260, 333, 84)
    H, P, mp = oracle.fill(a, b)
    out = engine.fill(a, b, h_dtype=torch.int64 if mode == "h64" else None, p_dtype=torch.int8 if mode.startswith("p8") else None,
                      want_h=mode in ("p8", "h64"), want_p=mode != "score_only")
    assert engine.get_option("last_strips2") == 10
    r = out.result()
    assert r["max_pos"] == mp and r["max_score"] == int(H.flat[mp])
    if out.H is not None:
        assert np.array_equal(out.H.cpu().numpy().astype(np.int32), H)
    if out.P is not None:
        assert np.array_equal(out.P.cpu().numpy().astype(np.int32), P)


@pytest.mark.parametrize("scores", [(5, -3, -4), (3, -3, 0), (2, 1, -3), (1, -1, -1)], ids=str)
def test_other_scorings(engine, oracle, scores):
    a, b = oracle.generate(700, 400, 79)
    check_against_oracle(engine, oracle, a, b, scores)
    assert engine.get_option("last_strips2") > 0


def test_same_results_as_the_one_column_kernel_and_more_passes_than_cus(engine, oracle):
    import torch
    a, b = oracle.generate(9000, 1600, 80)          # 72 strips
    engine.set_option("max_blocks", 7)              # 11 passes of the strip loop per workgroup
    try:
        two = engine.fill(a, b)
        assert engine.get_option("last_strips2") == 72
        engine.set_option("debug_flags", 16384)
        one = engine.fill(a, b)
        assert engine.get_option("last_strips2") == 0
    finally:
        engine.set_option("debug_flags", 0)
        engine.set_option("max_blocks", 0)
    assert torch.equal(two.H, one.H) and torch.equal(two.P, one.P) and two.result() == one.result()
    H, P, mp = oracle.fill(a, b)
    assert np.array_equal(two.H.cpu().numpy(), H) and np.array_equal(two.P.cpu().numpy(), P) and two.result()["max_pos"] == mp


def test_other_alphabets_and_ties(engine, oracle):
    rng = np.random.default_rng(5)
    for letters in (b"AC", b"ACGTN", b"xyzwvut"):
        a = rng.choice(np.frombuffer(letters, np.uint8), size=900).astype(np.uint8)
        b = rng.choice(np.frombuffer(letters, np.uint8), size=320).astype(np.uint8)
        check_against_oracle(engine, oracle, a, b)
        assert engine.get_option("last_strips2") > 0
    a = np.frombuffer(b"ACGT" * 200, np.uint8).copy()      # periodic: many cells share the maximum, the lowest index must win
    check_against_oracle(engine, oracle, a, a[:320].copy())
    eight = rng.choice(np.frombuffer(b"ABCDEFGH", np.uint8), size=640).astype(np.uint8)   # 8 letters: the one-column kernel's job
    check_against_oracle(engine, oracle, eight, eight[:160].copy())
    assert engine.get_option("last_strips2") > 0      # (launched, and left at once: the alphabet is only known on the device)


def test_more_strips_than_cus_streaming_checksums(engine, oracle, swamd):
    """40000 x 4096: 318 strips of 126 columns, i.e. a second pass of the strip loop on a 256-CU device -- or two column tiles; per-row
    checksums and arg-max against the streaming oracle."""
    a, b = swamd.generate(40000, 4096, 3)
    st = oracle.fill_streaming(a, b)
    for flags, tiles, strips in ((524288, 1, 318), (0, 2, 159)):   # 126-column strips: untiled (debug bit 19), then as two column tiles
        engine.set_option("debug_flags", flags); engine.set_option("s2w", 126)
        try:
            out = engine.fill(a, b)
        finally:
            engine.set_option("debug_flags", 0); engine.set_option("s2w", 0)
        assert engine.get_option("last_tiles") == tiles and engine.get_option("last_strips2") == strips
        r = out.result()
        assert r["max_pos"] == st["max_pos"] and r["max_score"] == st["max_score"]
        assert np.array_equal(engine.row_checksums(out.H), st["csH"]) and np.array_equal(engine.row_checksums(out.P), st["csP"])
        assert np.array_equal(out.H[-1].cpu().numpy(), st["bottom"])


@pytest.mark.parametrize("mode", ["p8", "p8_only", "p32_only", "h_only", "score_only"])
@pytest.mark.parametrize("cols,rows", [(1000, 704), (2520, 320), (126, 16)])
def test_output_modes(engine, oracle, cols, rows, mode):
    """int8 P, either matrix left out, score only: exact arg-max in every mode (without H the block statement records the cell)."""
    import torch
    a, b = oracle.generate(cols, rows, 81)
    H, P, mp = oracle.fill(a, b)
    want_h = mode in ("p8", "h_only")
    want_p = mode in ("p8", "p8_only", "p32_only")
    out = engine.fill(a, b, p_dtype=torch.int8 if mode.startswith("p8") else None, want_h=want_h, want_p=want_p)
    assert engine.get_option("last_strips2") == (cols + 125) // 126
    r = out.result()
    assert r["max_pos"] == mp and r["max_score"] == int(H.flat[mp])
    if want_h:
        assert np.array_equal(out.H.cpu().numpy(), H)
    if want_p:
        assert np.array_equal(out.P.cpu().numpy().astype(np.int32), P)
        path = engine.traceback(out, mp)
        P1 = P.copy()
        assert np.array_equal(path, oracle.backtrack(P1, mp)) and np.array_equal(out.P.cpu().numpy().astype(np.int32), P1)


def test_argmax_ties_without_h_two_columns(engine, oracle):
    for a, b in ((b"ACGT" * 80, b"ACGT" * 72), (b"A" * 300, b"A" * 208), (b"AC" * 150, b"CA" * 104)):
        H, P, mp = oracle.fill(a, b)
        r = engine.fill(a, b, want_h=False, want_p=False).result()
        assert engine.get_option("last_strips2") > 0
        assert r["max_pos"] == mp and r["max_score"] == int(H.flat[mp])


@pytest.mark.parametrize("cols,rows,cuts,p8,want_h", [(1000, 640, (320,), False, True), (4200, 1296, (416, 880), False, True),
                                                     (1500, 912, (304, 608), True, False), (252, 96, (16, 32, 48, 64), True, True)])
def test_band_resident_launches_on_the_two_column_kernel(engine, oracle, swamd, cols, rows, cuts, p8, want_h):
    """Stacked bands whose heights are multiples of 16: halo row in and last row out as granules, per-63-column-strip flags."""
    from test_band_gpu import _bands
    _bands(engine, oracle, swamd, cols, rows, cuts, p8=p8, want_h=want_h)
    assert engine.get_option("last_strips2") == (cols + 125) // 126


def test_lower_band_first_on_the_two_column_kernel(engine, oracle, swamd):
    from test_band_gpu import _bands
    ncu = engine.get_option("num_cus")
    _bands(engine, oracle, swamd, 5000, 2016, (1008,), reverse=True, max_blocks=max(8, ncu // 2 - 8))


@pytest.mark.parametrize("p8", [False, True], ids=["p32", "p8"])
@pytest.mark.parametrize("cols,rows", [(1000, 704), (2520, 320), (126, 16), (5000, 1296)])
def test_int64_h(engine, oracle, cols, rows, p8):
    """int64 H (BASELINE config 3's element type): {hA, 0, hB, 0} as one 16-byte store per lane and row."""
    import torch
    a, b = oracle.generate(cols, rows, 82)
    H, P, mp = oracle.fill(a, b)
    out = engine.fill(a, b, h_dtype=torch.int64, p_dtype=torch.int8 if p8 else None)
    assert engine.get_option("last_strips2") == (cols + 125) // 126
    assert out.H.dtype == torch.int64 and np.array_equal(out.H.cpu().numpy(), H.astype(np.int64))
    assert np.array_equal(out.P.cpu().numpy().astype(np.int32), P)
    r = out.result()
    assert r["max_pos"] == mp and r["max_score"] == int(H.flat[mp])
    assert np.array_equal(engine.row_checksums(out.H), oracle.row_checksums(H))


def test_int64_h_odd_columns_stay_on_the_one_column_kernel(engine, oracle):
    import torch
    a, b = oracle.generate(1007, 304, 83)
    H, P, mp = oracle.fill(a, b)
    out = engine.fill(a, b, h_dtype=torch.int64)
    assert engine.get_option("last_strips2") == 0
    assert np.array_equal(out.H.cpu().numpy(), H.astype(np.int64)) and np.array_equal(out.P.cpu().numpy(), P)
