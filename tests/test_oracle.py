"""CPU: the oracle (oracle/sw_oracle.c) against fixtures generated from the real reference."""
import glob
import os

import numpy as np
import pytest

from oracle_lib import GOLDEN, golden, golden_hashes

FULL = sorted(os.path.basename(f)[:-4] for f in glob.glob(os.path.join(GOLDEN, "*.npz")))


@pytest.mark.parametrize("name", FULL)
def test_fill_matches_reference(oracle, name):
    g = golden(name)
    H, P, mp = oracle.fill(g["a"], g["b"])
    assert np.array_equal(H, g["H"]) and np.array_equal(P, g["P0"])
    assert mp == int(g["meta"][3]) and int(H.flat[mp]) == int(g["meta"][4])
    Hw, Pw, mpw = oracle.fill(g["a"], g["b"], wavefront=True)
    assert np.array_equal(Hw, H) and np.array_equal(Pw, P) and mpw == mp


@pytest.mark.parametrize("name", FULL)
def test_backtrack_matches_reference(oracle, name):
    g = golden(name)
    P = g["P0"].copy()
    path = oracle.backtrack(P, int(g["meta"][3]))
    assert np.array_equal(P, g["P1"]) and np.array_equal(path, g["path"])
    assert len(path) == int(g["meta"][5])


def test_builtin_known_answers(oracle):
    # the reference's own asserts: serial_smithW.c:162-166, omp_smithW-v1-refinedOrig.cpp:229-238
    H, P, mp = oracle.fill("TGTTACGG", "GGTTGACTA")
    assert H.flat[H.size - 1] == 7 and mp == 69 and H.flat[69] == 13


def test_readme_screenshot_scoring(oracle):
    # Media/sampleOutput.png (orig scoring 5/-3/-4, omp_smithW_orig.c:65-67), SURVEY.md App. C KAT-2
    H, P, mp = oracle.fill("CTATCAA", "ACAGT", scores=(5, -3, -4))
    want = [[0, 0, 0, 5, 1, 0, 5, 5], [0, 5, 1, 1, 2, 6, 2, 2], [0, 1, 2, 6, 2, 2, 11, 7], [0, 0, 0, 2, 3, 0, 7, 8], [0, 0, 5, 1, 7, 3, 3, 4]]
    assert H[1:].tolist() == want and divmod(mp, 8) == (3, 6)


@pytest.mark.parametrize("name", sorted(golden_hashes()))
def test_generate_and_hashes(oracle, name):
    h = golden_hashes()[name]
    if name == "kat_builtin":
        return
    a, b = oracle.generate(h["cols"], h["rows"], h["seed"])
    assert bytes(a[:32]).decode() == h["a_head"] and bytes(b[:32]).decode() == h["b_head"]
    H, P, mp = oracle.fill(a, b)
    assert mp == h["maxPos"] and int(H.flat[mp]) == h["maxScore"]
    assert f"{oracle.fnv(H):016x}" == h["fnvH"] and f"{oracle.fnv(P):016x}" == h["fnvP0"]
    assert f"{oracle.fnv(oracle.row_checksums(H)):016x}" == h["fnv_csH"]
    st = oracle.fill_streaming(a, b)
    assert st["max_pos"] == mp and st["max_score"] == h["maxScore"]
    assert f"{oracle.fnv(st['csH']):016x}" == h["fnv_csH"] and f"{oracle.fnv(st['csP']):016x}" == h["fnv_csP"]
    assert np.array_equal(st["bottom"], H[-1])
    path = oracle.backtrack(P, mp)
    assert len(path) == h["pathLen"] and f"{oracle.fnv(P):016x}" == h["fnvP1"]
    assert (int(path.min()) if len(path) else -1) == h["path_min"]


def test_glibc_rand_first_draws(oracle):
    # first rand() outputs for seed 1
    import ctypes

    class Rng(ctypes.Structure):
        _fields_ = [("r", ctypes.c_uint32 * 34), ("k", ctypes.c_int)]

    g = Rng()
    oracle.L.swo_srand(ctypes.byref(g), 1)
    oracle.L.swo_rand.restype = ctypes.c_int32
    got = [oracle.L.swo_rand(ctypes.byref(g)) for _ in range(5)]
    assert got == [1804289383, 846930886, 1681692777, 1714636915, 1957747793]


def test_wavefront_indexing(oracle):
    # omp_smithW.c:260-291: every cell is visited exactly once, all cells of diagonal i have row+col == i+1
    for (m, n) in [(9, 10), (10, 9), (2, 2), (2, 7), (7, 2), (33, 33)]:
        seen = np.zeros((n, m), int)
        for i in range(1, m + n - 3 + 1):
            ne = oracle.n_element(i, m, n)
            si, sj = oracle.first_diag_element(i, m, n)
            for j in range(ne):
                assert si - j + sj + j == i + 1
                seen[si - j, sj + j] += 1
        assert (seen[1:, 1:] == 1).all() and seen[0].sum() == 0 and seen[:, 0].sum() == 0
