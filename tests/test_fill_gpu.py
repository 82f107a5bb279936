"""GPU: the HIP fill (through the C-ABI) against the golden fixtures and the oracle. Bit-exact."""
import glob
import os

import numpy as np
import pytest

from oracle_lib import GOLDEN, golden, golden_hashes

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=[0, 1], ids=["systolic", "strip_scan"])
def engine_kind(request, engine):
    """Every parity test runs against both fill engines behind the same C-ABI."""
    engine.set_option("engine", request.param)
    yield request.param
    engine.set_option("engine", 0)
FULL = sorted(os.path.basename(f)[:-4] for f in glob.glob(os.path.join(GOLDEN, "*.npz")))


def check_against_oracle(engine, oracle, a, b, scores=(3, -3, -2), h_dtype=None):
    out = engine.fill(a, b, scores, h_dtype=h_dtype)
    H, P, mp = oracle.fill(a, b, scores)
    dH = out.H.cpu().numpy()
    if not np.array_equal(dH.astype(np.int64), H.astype(np.int64)):
        bad = np.argwhere(dH.astype(np.int64) != H.astype(np.int64))
        r, c = bad[0]
        raise AssertionError(f"H differs in {len(bad)} cells, first at row {r} col {c} (strip {(c - 1) // 63}); rows {bad[:, 0].min()}..{bad[:, 0].max()}, "
                             f"cols {bad[:, 1].min()}..{bad[:, 1].max()}; got {dH[r, max(0, c - 2):c + 3].tolist()} want {H[r, max(0, c - 2):c + 3].tolist()}")
    assert np.array_equal(out.P.cpu().numpy(), P), "P differs"
    r = out.result()
    assert r["max_pos"] == mp and r["max_score"] == int(H.flat[mp])
    path = engine.traceback(out, mp)
    opath = oracle.backtrack(P, mp)
    assert np.array_equal(path, opath) and np.array_equal(out.P.cpu().numpy(), P)
    return out


@pytest.mark.parametrize("name", FULL)
def test_golden_fixtures(engine, name):
    g = golden(name)
    out = engine.fill(g["a"], g["b"])
    for what, got, want in (("H", out.H.cpu().numpy(), g["H"]), ("P", out.P.cpu().numpy(), g["P0"])):
        if not np.array_equal(got, want):
            bad = np.argwhere(got != want)
            r, c = bad[0]
            raise AssertionError(f"{what} differs in {len(bad)} cells: rows {bad[:, 0].min()}..{bad[:, 0].max()}, cols {bad[:, 1].min()}..{bad[:, 1].max()}; "
                                 f"first ({r},{c}) got {got[r, max(0, c - 2):c + 3].tolist()} want {want[r, max(0, c - 2):c + 3].tolist()}")
    r = out.result()
    assert r["max_pos"] == int(g["meta"][3]) and r["max_score"] == int(g["meta"][4])
    path = engine.traceback(out)
    assert np.array_equal(path, g["path"]) and np.array_equal(out.P.cpu().numpy(), g["P1"])


def test_builtin_known_answers(engine):
    # serial_smithW.c:162-166, omp_smithW-v1-refinedOrig.cpp:229-238
    out = engine.fill("TGTTACGG", "GGTTGACTA")
    H = out.H.cpu().numpy()
    assert H.flat[H.size - 1] == 7 and out.result() == {"max_pos": 69, "max_score": 13, "path_len": 0}


@pytest.mark.parametrize("name", ["rand_1024x1024_s1", "rand_4096x4096_s1", "rand_1000x3000_s5", "rand_3000x1000_s5", "rand_2049x2047_s9"])
def test_golden_hashes(engine, oracle, swamd, name):
    h = golden_hashes()[name]
    a, b = swamd.generate(h["cols"], h["rows"], h["seed"])
    out = engine.fill(a, b)
    r = out.result()
    assert r["max_pos"] == h["maxPos"] and r["max_score"] == h["maxScore"]
    assert f"{oracle.fnv(out.H.cpu().numpy()):016x}" == h["fnvH"]
    assert f"{oracle.fnv(out.P.cpu().numpy()):016x}" == h["fnvP0"]
    assert f"{oracle.fnv(engine.row_checksums(out.H)):016x}" == h["fnv_csH"]
    assert f"{oracle.fnv(engine.row_checksums(out.P)):016x}" == h["fnv_csP"]
    path = engine.traceback(out)
    assert len(path) == h["pathLen"] and f"{oracle.fnv(out.P.cpu().numpy()):016x}" == h["fnvP1"]


@pytest.mark.parametrize("cols,rows", [(1, 1), (1, 2), (2, 1), (63, 15), (64, 16), (65, 17), (127, 31), (128, 32), (129, 33),
                                       (1, 200), (200, 1), (500, 37), (37, 500), (640, 480), (1025, 1023)])
def test_ragged_sizes_vs_oracle(engine, oracle, cols, rows):
    rng = np.random.default_rng(cols * 7919 + rows)
    a = rng.integers(0, 4, cols).astype(np.uint8) + 65
    b = rng.integers(0, 4, rows).astype(np.uint8) + 65
    check_against_oracle(engine, oracle, a, b)


def test_degenerate_inputs(engine, oracle):
    check_against_oracle(engine, oracle, b"A" * 300, b"A" * 200)       # all match: one long diagonal
    check_against_oracle(engine, oracle, b"A" * 300, b"C" * 200)       # all mismatch: H == 0, maxPos 0, empty path
    check_against_oracle(engine, oracle, b"ACGT" * 80, b"ACGT" * 70)   # periodic: many arg-max ties
    check_against_oracle(engine, oracle, bytes(range(256)), bytes(range(255, -1, -1)))  # arbitrary bytes


def test_empty_sequences(engine):
    out = engine.fill(b"", b"ACGT")
    assert out.H.shape == (5, 1) and int(out.H.abs().sum()) == 0 and out.result()["max_pos"] == 0
    out = engine.fill(b"ACGT", b"")
    assert out.H.shape == (1, 5) and int(out.P.abs().sum()) == 0


@pytest.mark.parametrize("scores", [(5, -3, -4), (1, -1, -1), (2, -7, 0), (3, 1, -2)])
def test_other_scores(engine, oracle, scores):
    a, b = oracle.generate(333, 222, 4)
    check_against_oracle(engine, oracle, a, b, scores)


def test_readme_screenshot_scoring(engine):
    out = engine.fill("CTATCAA", "ACAGT", scores=(5, -3, -4))
    assert out.H.cpu().numpy()[3].tolist() == [0, 1, 2, 6, 2, 2, 11, 7] and divmod(out.result()["max_pos"], 8) == (3, 6)


def test_int64_h(engine, oracle):
    import torch
    a, b = oracle.generate(700, 300, 2)
    out = check_against_oracle(engine, oracle, a, b, h_dtype=torch.int64)
    assert out.H.dtype == torch.int64


def test_row_band_with_top_halo(engine, oracle):
    """Two stacked bands (the multi-GPU decomposition) reproduce the single fill."""
    a, b = oracle.generate(777, 400, 6)
    H, P, mp = oracle.fill(a, b)
    cut = 150
    top_out = engine.fill(a, b[:cut])
    assert np.array_equal(top_out.H.cpu().numpy(), H[: cut + 1])
    bot = engine.fill(a, b[cut:], top=H[cut])
    assert np.array_equal(bot.H.cpu().numpy(), H[cut:]) and np.array_equal(bot.P.cpu().numpy()[1:], P[cut + 1:])


def test_repeat_calls_are_deterministic(engine, oracle):
    a, b = oracle.generate(2000, 1500, 8)
    H, P, mp = oracle.fill(a, b)
    for _ in range(3):
        out = engine.fill(a, b)
        assert np.array_equal(out.H.cpu().numpy(), H) and out.result()["max_pos"] == mp


@pytest.mark.parametrize("wpb,maxb", [(1, 0), (8, 0), (4, 3)])
def test_launch_shapes(engine, oracle, wpb, maxb):
    """Different waves-per-workgroup / capped grids (several strips per wave) give the same matrices."""
    a, b = oracle.generate(3000, 600, 12)
    H, P, mp = oracle.fill(a, b)
    engine.set_option("waves_per_block", wpb)
    engine.set_option("max_blocks", maxb)
    try:
        out = engine.fill(a, b)
        assert np.array_equal(out.H.cpu().numpy(), H) and np.array_equal(out.P.cpu().numpy(), P)
        assert out.result()["max_pos"] == mp
    finally:
        engine.set_option("waves_per_block", 4)
        engine.set_option("max_blocks", 0)


def test_full_size_16384_streaming_checksums(engine, oracle, swamd):
    """BASELINE config 2 (16384 x 16384 int32): per-row checksums + arg-max vs the streaming oracle."""
    a, b = swamd.generate(16384, 16384, 1)
    st = oracle.fill_streaming(a, b)
    out = engine.fill(a, b)
    r = out.result()
    assert r["max_pos"] == st["max_pos"] and r["max_score"] == st["max_score"]
    assert np.array_equal(engine.row_checksums(out.H), st["csH"])
    assert np.array_equal(engine.row_checksums(out.P), st["csP"])
    assert np.array_equal(out.H[-1].cpu().numpy(), st["bottom"])
    n = engine.traceback(out, want_path=False)
    assert 16384 < n < 3 * 16384
    if engine.get_option("engine") == 0:
        # the same workload with the compact predecessor matrix and with the other workgroup shape / store policy:
        # identical checksums (an int8 P checksums like its int32 widening), arg-max and path length
        import torch
        for opts, p_dtype in (({}, torch.int8), ({"strips_per_group": 2, "consumers": 4, "store_policy": 1}, None)):
            for k, v in opts.items():
                engine.set_option(k, v)
            try:
                o2 = engine.fill(a, b, p_dtype=p_dtype)
            finally:
                for k in opts:
                    engine.set_option(k, 0)
            r2 = o2.result()
            assert r2["max_pos"] == st["max_pos"] and r2["max_score"] == st["max_score"]
            assert np.array_equal(engine.row_checksums(o2.H), st["csH"]) and np.array_equal(engine.row_checksums(o2.P), st["csP"])
            assert engine.traceback(o2, want_path=False) == n
            del o2
