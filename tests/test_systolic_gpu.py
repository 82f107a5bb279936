"""GPU: systolic-engine-only parity tests (tuning options, compact P, BASELINE config 3 at full size)."""
import numpy as np
import pytest

from test_fill_gpu import check_against_oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def engine_kind(engine):
    engine.set_option("engine", 0)
    yield 0


TUNINGS = [
    {"strips_per_group": 2, "consumers": 2}, {"strips_per_group": 2, "consumers": 3}, {"strips_per_group": 2, "consumers": 4}, {"strips_per_group": 1, "consumers": 2},
    {"strips_per_group": 1, "consumers": 3}, {"strips_per_group": 1, "consumers": 4}, {"strips_per_group": 1, "consumers": 6},
    {"strips_per_group": 1, "consumers": 8}, {"store_policy": 1}, {"store_policy": 2}, {"xcd_order": 1},
    {"xcd_order": 1, "max_blocks": 20}, {"pace_ps": 40000}, {"importers": 2}, {"importers": 1}, {"importers": 4, "consumers": 6}, {"importers": 6, "strips_per_group": 1, "consumers": 2}, {"importers": 2, "strips_per_group": 1, "consumers": 4}, {"store_policy": 2, "xcd_order": 1, "strips_per_group": 1, "consumers": 8},
]


@pytest.mark.parametrize("opts", TUNINGS, ids=lambda o: ",".join(f"{k}={v}" for k, v in o.items()))
@pytest.mark.parametrize("h64", [False, True], ids=["h32", "h64"])
def test_tuning_options_do_not_change_results(engine, oracle, engine_kind, opts, h64):
    """Workgroup shape, store cache policy, XCD order and pacing only move time: H, P and the arg-max stay bit-exact."""
    import torch
    a, b = oracle.generate(4200, 1300, 21)   # 67 strips: enough groups for the XCD-aware order and a second pass
    defaults = {k: engine.get_option(k) for k in opts}
    for k, v in opts.items():
        engine.set_option(k, v)
    try:
        check_against_oracle(engine, oracle, a, b, h_dtype=torch.int64 if h64 else None)
    finally:
        for k, v in defaults.items():
            engine.set_option(k, v)


@pytest.mark.parametrize("h64", [False, True], ids=["h32", "h64"])
@pytest.mark.parametrize("cols,rows", [(1000, 700), (63, 16), (4200, 1300), (130, 1029)])
def test_compact_p_int8(engine, oracle, swamd, engine_kind, cols, rows, h64):
    """sw_fill_device_ex with one byte per predecessor code: same H, arg-max, codes, traceback and checksums."""
    import torch
    a, b = oracle.generate(cols, rows, 29)
    H, P, mp = oracle.fill(a, b)
    for policy in (1, 2):
        engine.set_option("store_policy", policy)
        try:
            out = engine.fill(a, b, h_dtype=torch.int64 if h64 else None, p_dtype=torch.int8)
        finally:
            engine.set_option("store_policy", 0)
        assert out.P.dtype == torch.int8
        assert np.array_equal(out.H.cpu().numpy().astype(np.int64), H.astype(np.int64))
        assert np.array_equal(out.P.cpu().numpy().astype(np.int32), P)
        assert out.result()["max_pos"] == mp
        assert np.array_equal(engine.row_checksums(out.P), oracle.row_checksums(P))   # int8 P checksums like its int32 widening
    P8 = out.P.cpu().numpy().copy()
    path = engine.traceback(out, mp)
    P1 = P.copy()
    opath = oracle.backtrack(P1, mp)
    assert np.array_equal(path, opath) and np.array_equal(out.P.cpu().numpy().astype(np.int32), P1)
    hpath = swamd.traceback_host(P8, mp)
    assert np.array_equal(hpath, opath) and np.array_equal(P8.astype(np.int32), P1)
    # round trip to the reference's int32 layout on the device (sw_p8_to_p32_device), negated path included
    wide = engine.widen_p(out.P)
    engine.synchronize()
    assert wide.dtype == torch.int32 and np.array_equal(wide.cpu().numpy(), P1)
    odd = engine.widen_p(out.P.view(-1)[3:-2])        # unaligned start and a ragged tail
    engine.synchronize()
    assert np.array_equal(odd.cpu().numpy(), P1.reshape(-1)[3:-2])


def test_config3_65536_int64_vs_oracle_digest(engine, oracle, swamd, engine_kind):
    """BASELINE config 3 (65536 x 65536, int64 H + int32 P, resident in HBM) against the streaming oracle's whole-matrix digest
    (tests/golden/big_digests.json): per-row checksums of H and P, arg-max, the bottom row, the traced-back path (every index) and P
    after the traceback; every int64 H must be the sign extension of its int32 value (sw_row_checksums_device checks that)."""
    import json, os, torch
    from oracle_lib import GOLDEN
    free, _ = torch.cuda.mem_get_info()
    if free < (60 << 30):
        pytest.skip(f"needs 60 GB of free HBM, {free >> 30} GB free")
    D = json.load(open(os.path.join(GOLDEN, "big_digests.json")))["rand_65536x65536_s1"]
    fnv = lambda x: f"{oracle.fnv(np.ascontiguousarray(x)):016x}"
    n = 65536
    a, b = swamd.generate(n, n, 1)
    d_a, _ = engine.to_device(a)
    d_b, _ = engine.to_device(b)
    out = engine.alloc(n, n, torch.int64)
    engine.fill_into(out, d_a, d_b)
    engine.synchronize()
    r = out.result()
    csH, csP = engine.row_checksums(out.H), engine.row_checksums(out.P)
    bottom = out.H[-1].cpu().numpy()
    assert r["max_pos"] == D["maxPos"] and r["max_score"] == D["maxScore"]
    assert fnv(csH) == D["fnv_csH"] and fnv(csP) == D["fnv_csP"]
    for d in D["bands"]:
        assert fnv(csH[d["lo"] + 1:d["hi"] + 1]) == d["fnv_csH"] and fnv(csP[d["lo"] + 1:d["hi"] + 1]) == d["fnv_csP"]
    assert bottom.dtype == np.int64 and fnv(bottom.astype(np.int32)) == D["bands"][-1]["fnv_bottom_H"]
    path = engine.traceback(out)
    assert len(path) == D["pathLen"] and fnv(path) == D["fnv_path"] and int(path[-1]) == D["path_end"]
    assert fnv(engine.row_checksums(out.P)) == D["fnv_csP1"]


def test_strip_scan_engine_rejects_compact_p(engine, oracle, swamd):
    import torch
    a, b = oracle.generate(200, 100, 3)
    engine.set_option("engine", 1)
    try:
        with pytest.raises(swamd.SwError):
            engine.fill(a, b, p_dtype=torch.int8)
    finally:
        engine.set_option("engine", 0)


@pytest.mark.parametrize("cols,rows", [(1000, 700), (63, 16), (4200, 1300)])
@pytest.mark.parametrize("mode", ["p8_only", "p32_only", "h_only", "score_only"])
def test_matrix_less_fills(engine, oracle, mode, cols, rows):
    """sw_fill_device_ex with d_H and/or d_P NULL (P-only, H-only, score-only): what is written is
    bit-exact, the arg-max is exact in every mode (serial_smithW.c:240-242), the P-only traceback equals backtrack()."""
    import torch
    a, b = oracle.generate(cols, rows, 31)
    H, P, mp = oracle.fill(a, b)
    want_h = mode == "h_only"
    want_p = mode in ("p8_only", "p32_only")
    out = engine.fill(a, b, p_dtype=torch.int8 if mode == "p8_only" else None, want_h=want_h, want_p=want_p)
    r = out.result()
    assert r["max_pos"] == mp and r["max_score"] == int(H.flat[mp])
    assert (out.H is None) == (not want_h) and (out.P is None) == (not want_p)
    if want_h:
        assert np.array_equal(out.H.cpu().numpy(), H)
    if want_p:
        assert np.array_equal(out.P.cpu().numpy().astype(np.int32), P)
        path = engine.traceback(out, mp)
        P1 = P.copy()
        assert np.array_equal(path, oracle.backtrack(P1, mp)) and np.array_equal(out.P.cpu().numpy().astype(np.int32), P1)


def test_argmax_ties_without_h(engine, oracle):
    """Score-only arg-max on inputs full of ties: the LOWEST linear index among the maxima (serial_smithW.c:240-242)."""
    for a, b in ((b"ACGT" * 80, b"ACGT" * 70), (b"A" * 300, b"A" * 200), (b"A" * 300, b"C" * 200), (b"ACGT" * 300, b"TGCA" * 290)):
        H, P, mp = oracle.fill(a, b)
        r = engine.fill(a, b, want_h=False, want_p=False).result()
        assert r["max_pos"] == mp and r["max_score"] == int(H.flat[mp])


def test_stalled_strip_times_out_and_context_recovers(engine, oracle, swamd):
    """The in-kernel abort path: a strip that never runs (debug bit 3) makes every wait give up; the call reports
    SW_ETIMEOUT instead of hanging, and the same context then fills correctly again."""
    a, b = oracle.generate(600, 300, 17)
    engine.set_option("debug_flags", 8)
    try:
        with pytest.raises(swamd.SwError) as e:
            engine.fill(a, b).result()
        assert e.value.code == -62
    finally:
        engine.set_option("debug_flags", 0)
    check_against_oracle(engine, oracle, a, b)


@pytest.mark.parametrize("nletters", [1, 2, 5, 7, 8, 20])
def test_alphabet_sizes_pick_the_producer(engine, oracle, nletters):
    """Up to 7 distinct letters run the perm producer (letter codes + v_perm profile), more fall back to the
    character-compare producer: same matrices either way."""
    rng = np.random.default_rng(100 + nletters)
    letters = rng.choice(np.arange(33, 127), size=nletters, replace=False).astype(np.uint8)
    a = letters[rng.integers(0, nletters, 1500)]
    b = letters[rng.integers(0, nletters, 900)]
    check_against_oracle(engine, oracle, a, b)
    check_against_oracle(engine, oracle, a, b, scores=(5, 2, -3))     # positive mismatch
    check_against_oracle(engine, oracle, a, b, scores=(100, -90, -20))  # scores beyond a signed byte: character-compare producer


@pytest.mark.parametrize("cols,rows", [(1000, 700), (63, 16), (4200, 1300), (130, 1029), (1, 1), (64, 200)])
def test_character_compare_producers_still_exact(engine, oracle, cols, rows):
    """debug bit 4 switches the perm producer off: the round-1 producers (fast / generic) behind the same tests."""
    a, b = oracle.generate(cols, rows, 33)
    engine.set_option("debug_flags", 16)
    try:
        check_against_oracle(engine, oracle, a, b)
        check_against_oracle(engine, oracle, a, b, scores=(3, 1, -2))
    finally:
        engine.set_option("debug_flags", 0)


def test_many_launches_of_mixed_shapes_across_the_tag_wrap(engine, swamd):
    """The perm producer's exported edge values carry an 8-bit launch tag in their top byte and the workspace is reused
    by launches of different shapes; 800 back-to-back fills of three golden problems (the tag wraps after 255) must all
    be bit-exact.  Round 2 regression: values computed below the matrix from left-over LDS contents once grew past 24
    bits, bumped their tag byte and were accepted by the next launch -- about one wrong fill in 150."""
    from oracle_lib import golden
    gs = [golden(n) for n in ("rand_256x256_s1", "rand_300x200_s1", "rand_129x64_s11")]
    for it in range(270):
        for g in gs:
            out = engine.fill(g["a"], g["b"])
            assert np.array_equal(out.H.cpu().numpy(), g["H"]) and np.array_equal(out.P.cpu().numpy(), g["P0"]), f"iteration {it}"
            assert out.result()["max_pos"] == int(g["meta"][3])


def test_align_auto_picks_host_or_gpu(engine, oracle, swamd):
    """sw_align_auto (adaptive dispatch, omp_smithW-v7-adaptive.cpp:304-396): tiny problems on the host, the rest on the
    GPU; either way the result equals serial_smithW including the negated path."""
    for cols, rows, want_gpu in ((40, 30, False), (300, 200, False), (1200, 900, True)):
        a, b = oracle.generate(cols, rows, 3)
        H, P, mp = oracle.fill(a, b)
        path = oracle.backtrack(P, mp)
        r = swamd.align_auto(a, b, engine=engine)
        assert r["used_gpu"] == want_gpu
        assert np.array_equal(r["H"], H) and np.array_equal(r["P"], P) and r["max_pos"] == mp and r["path_len"] == len(path)


@pytest.mark.parametrize("cols,rows,seed", [(8, 9, 1), (1000, 700, 2), (4097, 333, 3), (22000, 60, 4), (22001, 33, 5), (126 * 171, 17, 6)])
def test_fill_host_is_the_reference_fill_on_host_buffers(engine, oracle, swamd, cols, rows, seed):
    """sw_fill_host, the entry point of INTEGRATION.md section 2: host sequences in, H / P / maxPos in the reference's layout out -- narrow,
    odd and wide shapes (the wide ones run as overlapping strips with streaming stores, the copy-out follows on two streams)"""
    a, b = oracle.generate(cols, rows, seed)
    H, P, mp = oracle.fill(a, b)
    r = swamd.fill_host(engine, a, b)
    assert (r["max_pos"], r["max_score"]) == (mp, int(H.flat[mp]))
    assert np.array_equal(r["H"], H) and np.array_equal(r["P"], P)
