"""GPU: the one-pair-per-wave batch kernel (csrc/sw_batch.hip, BASELINE config 5) and the wave-cooperative traceback
(csrc/sw_traceback.hip) against the oracle: every column-per-lane variant, every output mode, ragged and multi-strip shapes,
alphabets up to and beyond the 8 letters of the profile look-up, unusual scorings, long paths that cross many windows."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _pairs(rng, npairs, cols, rows, letters=4):
    alpha = np.frombuffer(b"ACGTNRYKMSWBDHVU", np.uint8)[:letters]
    A = alpha[rng.integers(0, letters, (npairs, cols))]
    B = alpha[rng.integers(0, letters, (npairs, rows))]
    return np.ascontiguousarray(A), np.ascontiguousarray(B)


def _check(engine, oracle, A, B, scores=(3, -3, -2), p_dtype=None, store_h=True, expect_wave=True):
    import torch
    res, H, P = engine.batch(A, B, scores=scores, store=True, p_dtype=p_dtype, store_h=store_h)
    # (an int8 P as the only matrix of pairs wider than 512 columns whose scores fit 12 bits: two pairs per wave on packed lanes)
    packed_p = p_dtype is not None and not store_h and A.shape[1] > 512 and A.shape[0] >= 2 and scores[0] * min(A.shape[1], B.shape[1]) < 32000 and -scores[2] < 32000
    assert engine.get_option("last_batch_kernel") == ((2 if packed_p else 1) if expect_wave else 0)
    res = res.cpu().numpy()
    for k in range(A.shape[0]):
        h, p, mp = oracle.fill(A[k], B[k], scores)
        if H is not None:
            assert np.array_equal(H[k].cpu().numpy(), h), f"pair {k} H"
        assert np.array_equal(P[k].cpu().numpy().astype(np.int32), p), f"pair {k} P"
        assert res[k, 0] == mp and res[k, 1] == int(h.flat[mp]), f"pair {k} arg-max {res[k]} vs {mp}"
    res2, _, _ = engine.batch(A, B, scores=scores, store=False)
    assert np.array_equal(res2.cpu().numpy(), res), "score-only mode"


@pytest.mark.parametrize("cols,rows", [(1, 1), (3, 70), (64, 5), (130, 70), (255, 33), (256, 256), (257, 100), (300, 17), (512, 64), (513, 80),
                                       (777, 129), (1024, 40), (1023, 65)])
def test_shapes_int32(engine, oracle, cols, rows):
    """4, 8 and 16 columns per lane; full and ragged last lanes; rows from one to more than the wave skew."""
    A, B = _pairs(np.random.default_rng(cols * 1000 + rows), 9, cols, rows)
    _check(engine, oracle, A, B)


@pytest.mark.parametrize("cols,rows", [(130, 70), (256, 90), (300, 50), (512, 31), (1000, 77), (1024, 64)])
def test_shapes_int8_p_without_h(engine, oracle, cols, rows):
    import torch
    A, B = _pairs(np.random.default_rng(cols + rows), 7, cols, rows)
    _check(engine, oracle, A, B, p_dtype=torch.int8, store_h=False)
    _check(engine, oracle, A, B, p_dtype=torch.int8, store_h=True)


@pytest.mark.parametrize("cols,rows", [(1025, 50), (1500, 300), (2048, 33), (2500, 200)])
def test_wider_than_one_strip(engine, oracle, cols, rows):
    """More than 64 x 16 columns: the wave sweeps strip after strip, the boundary column goes through the scratch row."""
    import torch
    A, B = _pairs(np.random.default_rng(cols), 5, cols, rows)
    _check(engine, oracle, A, B)
    _check(engine, oracle, A, B, p_dtype=torch.int8, store_h=False)


@pytest.mark.parametrize("letters", [1, 2, 3, 5, 7, 8])
def test_alphabets_up_to_8_letters(engine, oracle, letters):
    A, B = _pairs(np.random.default_rng(letters), 6, 200, 150, letters)
    _check(engine, oracle, A, B)


def test_alphabet_of_more_than_8_letters_takes_the_other_path(engine, oracle):
    A, B = _pairs(np.random.default_rng(9), 5, 200, 150, 12)
    _check(engine, oracle, A, B, expect_wave=False)


@pytest.mark.parametrize("scores", [(5, -3, -4), (1, -1, -1), (2, 1, -1), (3, -3, 0), (10, -20, -7), (0, 0, 0)])
def test_scorings(engine, oracle, scores):
    """other match / mismatch / gap values, a positive mismatch score and a zero gap included"""
    A, B = _pairs(np.random.default_rng(abs(hash(scores)) % 1000), 6, 300, 120)
    _check(engine, oracle, A, B, scores=scores)


def test_ties_and_empty_alignments(engine, oracle):
    """periodic sequences (many cells share the maximum: lowest linear index wins), all-match, no match at all"""
    cols, rows = 400, 300
    A = np.tile(np.frombuffer(b"ACGT", np.uint8), (6, cols // 4))
    B = np.tile(np.frombuffer(b"ACGT", np.uint8), (6, rows // 4))
    A[1], B[1] = 65, 65
    A[2], B[2] = 65, 67
    A[3] = np.frombuffer(b"AC", np.uint8).repeat(cols // 2)
    B[4] = np.frombuffer(b"GT", np.uint8).repeat(rows // 2)
    _check(engine, oracle, A, B)


def test_batch_traceback_of_every_pair(engine, oracle):
    import torch
    A, B = _pairs(np.random.default_rng(77), 300, 333, 222)
    A[3], B[3] = 65, 67
    for p_dtype in (torch.int8, None):
        res, H, P, paths = engine.batch(A, B, store=True, p_dtype=p_dtype, store_h=False, traceback=True, want_paths=True)
        res, P, paths = res.cpu().numpy(), P.cpu().numpy(), paths.cpu().numpy()
        for k in range(A.shape[0]):
            h, p, mp = oracle.fill(A[k], B[k])
            opath = oracle.backtrack(p, mp)
            assert res[k, 0] == mp and res[k, 2] == len(opath), f"pair {k}"
            assert np.array_equal(paths[k, :len(opath)], opath) and np.array_equal(P[k].astype(np.int32), p), f"pair {k} path"


@pytest.mark.parametrize("shape", [(2000, 1500), (4096, 4096), (100, 3000), (3000, 100)])
def test_traceback_long_paths(engine, oracle, shape):
    """paths of thousands of steps across many 64 x 64 windows, int32 and int8 P; similar sequences give long diagonals,
    a mutated copy with insertions gives runs of UP / LEFT"""
    import torch
    cols, rows = shape
    rng = np.random.default_rng(cols)
    a = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, cols)]
    src = np.resize(a, rows).copy()
    mut = rng.random(rows) < 0.08
    src[mut] = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, int(mut.sum()))]
    keep = rng.random(rows) > 0.03                      # deletions shift the diagonal
    b = np.resize(src[keep], rows)
    for p_dtype in (None, torch.int8):
        out = engine.fill(a, b, p_dtype=p_dtype)
        H, P, mp = oracle.fill(a, b)
        assert out.result()["max_pos"] == mp
        path = engine.traceback(out, mp)
        opath = oracle.backtrack(P, mp)
        assert len(path) == len(opath) and np.array_equal(path, opath)
        assert np.array_equal(out.P.cpu().numpy().astype(np.int32), P)
        assert out.result()["path_len"] == len(opath)


# ---- two pairs per wave on packed 16-bit lanes (sw_batch_wave16: score-only batches of 513+ columns whose scores fit 15 bits) ----------
def _check16(engine, oracle, A, B, scores=(3, -3, -2), expect=2):
    res, _, _ = engine.batch(A, B, scores=scores, store=False)
    assert engine.get_option("last_batch_kernel") == expect
    res = res.cpu().numpy()
    for k in range(A.shape[0]):
        h, p, mp = oracle.fill(A[k], B[k], scores)
        assert res[k, 0] == mp and res[k, 1] == int(h.flat[mp]) and res[k, 2] == 0, f"pair {k}: {res[k]} vs maxPos {mp} score {int(h.flat[mp])}"


@pytest.mark.parametrize("cols,rows,npairs", [(513, 1, 2), (600, 70, 9), (1024, 300, 8), (1000, 129, 3), (777, 64, 17), (1023, 65, 2), (1024, 1024, 5)])
def test_packed16_shapes_even_and_odd_batches(engine, oracle, cols, rows, npairs):
    A, B = _pairs(np.random.default_rng(cols + 7 * rows), npairs, cols, rows)
    _check16(engine, oracle, A, B)


@pytest.mark.parametrize("cols,rows", [(1025, 50), (1500, 300), (2500, 200)])
def test_packed16_wider_than_one_strip(engine, oracle, cols, rows):
    A, B = _pairs(np.random.default_rng(cols), 5, cols, rows)
    _check16(engine, oracle, A, B)


@pytest.mark.parametrize("letters", [1, 2, 3, 4, 5, 7, 8])
def test_packed16_alphabets(engine, oracle, letters):
    """up to 4 letters: the half-size score profiles (LE4); 5..8: both halves"""
    A, B = _pairs(np.random.default_rng(letters), 6, 700, 150, letters)
    _check16(engine, oracle, A, B)


@pytest.mark.parametrize("scores", [(5, -3, -4), (1, -1, -1), (2, 1, -1), (3, -3, 0), (10, -20, -7), (0, 0, 0), (100, -100, -90)])
def test_packed16_scorings(engine, oracle, scores):
    A, B = _pairs(np.random.default_rng(abs(hash(scores)) % 1000), 6, 640, 120)
    _check16(engine, oracle, A, B, scores=scores)


def test_packed16_ties_identical_and_disjoint_pairs(engine, oracle):
    """periodic sequences (the maximum is attained in many cells of a row and of a lane: the lowest linear index wins), all-match (the
    highest scores the shape allows), no match at all, and two very different pairs sharing a wave"""
    cols, rows = 800, 300
    A = np.tile(np.frombuffer(b"ACGT", np.uint8), (8, cols // 4))
    B = np.tile(np.frombuffer(b"ACGT", np.uint8), (8, rows // 4))
    A[1], B[1] = 65, 65
    A[2], B[2] = 65, 67
    A[3] = np.frombuffer(b"AC", np.uint8).repeat(cols // 2)
    B[4] = np.frombuffer(b"GT", np.uint8).repeat(rows // 2)
    A[6], B[6] = 71, 71
    A[7], B[7] = 84, 67
    _check16(engine, oracle, A, B)


def test_packed16_falls_back_when_scores_need_more_than_15_bits(engine, oracle):
    A, B = _pairs(np.random.default_rng(5), 4, 1024, 900)
    A[1], B[1] = 65, 65                                  # all-match: 900 x 40 = 36000 > 2^15
    _check16(engine, oracle, A, B, scores=(40, -3, -2), expect=1)
    A, B = _pairs(np.random.default_rng(6), 1, 1024, 100)   # a single pair has nobody to share a wave with
    _check16(engine, oracle, A, B, expect=1)


@pytest.mark.parametrize("match,cols,rows,expect_k12", [(3, 1024, 1024, True), (3, 1365, 1365, True), (3, 1366, 1366, False), (4, 1024, 1024, False), (7, 600, 585, True), (7, 600, 586, False)])
def test_packed16_keyed_argmax_up_to_12_bit_scores(engine, oracle, match, cols, rows, expect_k12):
    """match x min(cols, rows) < 4096: the arg-max tree runs on score x 16 + column keys; at and above the limit on plain scores with a descent.
    Either way the same result -- the all-match pair reaches the highest score the shape allows (4095 at the edge), periodic pairs tie in
    many columns of a lane and many rows"""
    assert (match * min(cols, rows) < 4096) == expect_k12
    A, B = _pairs(np.random.default_rng(match * cols + rows), 6, cols, rows)
    A[1], B[1] = 65, 65                                   # all-match
    A[2] = np.resize(np.frombuffer(b"ACGT", np.uint8), cols); B[2] = np.resize(np.frombuffer(b"ACGT", np.uint8), rows)
    A[3] = np.resize(np.frombuffer(b"AACC", np.uint8), cols); B[3] = np.resize(np.frombuffer(b"AC", np.uint8), rows)
    _check16(engine, oracle, A, B, scores=(match, -3, -2))


# ---- packed lanes with an int8 P (sw_batch_wave16<.., PB1>): the P codes come out of packed arithmetic, two pairs per wave ---------------
def _check16p(engine, oracle, A, B, scores=(3, -3, -2), expect=2):
    import torch
    res, H, P = engine.batch(A, B, scores=scores, store=True, p_dtype=torch.int8, store_h=False)
    assert H is None and engine.get_option("last_batch_kernel") == expect
    res, P = res.cpu().numpy(), P.cpu().numpy()
    for k in range(A.shape[0]):
        h, p, mp = oracle.fill(A[k], B[k], scores)
        assert np.array_equal(P[k].astype(np.int32), p), f"pair {k} P: first difference at {np.argwhere(P[k].astype(np.int32) != p)[:3].tolist()}"
        assert res[k, 0] == mp and res[k, 1] == int(h.flat[mp]), f"pair {k} arg-max {res[k]} vs {mp}"


@pytest.mark.parametrize("cols,rows,npairs", [(513, 1, 2), (600, 70, 9), (1024, 300, 8), (1000, 129, 3), (777, 64, 17), (1023, 65, 2), (1024, 1024, 5), (528, 33, 4)])
def test_packed16_p8_shapes_even_and_odd_batches(engine, oracle, cols, rows, npairs):
    """full and ragged last lanes (cols % 16), rows from one to more than the wave skew, an odd last pair (runs alone in both halves, stored once)"""
    A, B = _pairs(np.random.default_rng(cols + 7 * rows), npairs, cols, rows)
    _check16p(engine, oracle, A, B)


@pytest.mark.parametrize("cols,rows", [(1025, 50), (1500, 300), (2500, 200)])
def test_packed16_p8_wider_than_one_strip(engine, oracle, cols, rows):
    A, B = _pairs(np.random.default_rng(cols), 5, cols, rows)
    _check16p(engine, oracle, A, B)


@pytest.mark.parametrize("letters", [1, 2, 3, 4, 5, 7, 8])
def test_packed16_p8_alphabets(engine, oracle, letters):
    A, B = _pairs(np.random.default_rng(letters), 6, 700, 150, letters)
    _check16p(engine, oracle, A, B)


@pytest.mark.parametrize("scores", [(5, -3, -4), (1, -1, -1), (2, 1, -1), (3, -3, 0), (6, -20, -7), (0, 0, 0)])
def test_packed16_p8_scorings(engine, oracle, scores):
    """every tie-break of the P code: a positive mismatch and a zero gap make DIAGONAL / UP / LEFT attain the maximum together"""
    A, B = _pairs(np.random.default_rng(abs(hash(scores)) % 1000), 6, 640, 120)
    _check16p(engine, oracle, A, B, scores=scores)


def test_packed16_p8_ties_identical_and_disjoint_pairs_and_traceback(engine, oracle):
    import torch
    cols, rows = 800, 300
    A = np.tile(np.frombuffer(b"ACGT", np.uint8), (8, cols // 4))
    B = np.tile(np.frombuffer(b"ACGT", np.uint8), (8, rows // 4))
    A[1], B[1] = 65, 65
    A[2], B[2] = 65, 67
    A[3] = np.frombuffer(b"AC", np.uint8).repeat(cols // 2)
    B[4] = np.frombuffer(b"GT", np.uint8).repeat(rows // 2)
    A[6], B[6] = 71, 71
    A[7], B[7] = 84, 67
    _check16p(engine, oracle, A, B)
    res, H, P, paths = engine.batch(A, B, store=True, p_dtype=torch.int8, store_h=False, traceback=True, want_paths=True)
    assert engine.get_option("last_batch_kernel") == 2
    res, paths = res.cpu().numpy(), paths.cpu().numpy()
    for k in range(A.shape[0]):
        h, p, mp = oracle.fill(A[k], B[k])
        opath = oracle.backtrack(p, mp)
        assert res[k, 0] == mp and res[k, 2] == len(opath) and np.array_equal(paths[k, :len(opath)], opath), f"pair {k}"


def test_packed16_p8_above_12_bit_scores_and_the_15_bit_fall_back(engine, oracle):
    A, B = _pairs(np.random.default_rng(5), 4, 1024, 900)
    A[1], B[1] = 65, 65
    _check16p(engine, oracle, A, B, scores=(5, -3, -2))               # 5 x 900 = 4500 > 4095: the arg-max by descent, still two pairs per wave
    _check16p(engine, oracle, A, B, scores=(40, -3, -2), expect=1)    # 40 x 900 = 36000 > 2^15: one pair per wave
