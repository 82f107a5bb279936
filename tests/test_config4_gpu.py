"""GPU: BASELINE config 4 (262144 x 262144, row bands over 8 GPUs) AT SIZE on one MI355X.

  * band 0 of the 8-way split at full width -- 262144 columns x 32768 rows, int32 H + int8 P, one band-resident launch
    (sw_fill_band_device) -- against the streaming oracle: per-row checksums of H and P, arg-max, the bottom row both as
    matrix row and as the granules the next band would poll;
  * the WHOLE 262144^2 matrix as int8 P only (69 GB: what one GPU can keep), two ways -- one monolithic launch and eight
    stacked band-resident launches through sw_multi_* (devices [0]*8: the relay, the chunking and the tags of the 8-GPU run,
    the bands sharing this GPU's CUs) -- which must agree row for row, in the arg-max and in the traceback; the first 32768
    rows also against the oracle.
The oracle's 8.6e9-cell streaming fill (~45 s) is shared by both tests."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
COLS, BAND = 262144, 32768


@pytest.fixture(scope="module")
def band0(oracle, swamd):
    a, b = swamd.generate(COLS, COLS, 1)
    st = oracle.fill_streaming(a, b[:BAND])
    return a, b, st


def _free_gb():
    import torch
    return torch.cuda.mem_get_info()[0] >> 30


def test_band0_of_config4_full_width_vs_streaming_oracle(engine, band0):
    import torch
    if _free_gb() < 60:
        pytest.skip(f"needs 60 GB of free HBM, {_free_gb()} GB free")
    a, b, st = band0
    d_a, _ = engine.to_device(a)
    d_b, _ = engine.to_device(b[:BAND])
    H = torch.empty((BAND + 1, COLS + 1), dtype=torch.int32, device="cuda")
    P = torch.empty((BAND + 1, COLS + 1), dtype=torch.int8, device="cuda")
    P[0].zero_()
    res = torch.zeros(3, dtype=torch.int64, device="cuda")
    S = (COLS + 62) // 63
    gran = torch.zeros(COLS + 1, dtype=torch.int64, device="cuda")
    done = torch.zeros(S, dtype=torch.int32, device="cuda")
    engine.fill_band(d_a, COLS, d_b, BAND, COLS, H, P, res, bot_gran=gran, bot_tag=5, bot_done=done)
    engine.synchronize()
    r = res.cpu().tolist()
    assert r[2] == 0
    assert (r[0], r[1]) == (st["max_pos"], st["max_score"])
    assert np.array_equal(engine.row_checksums(H), st["csH"])
    assert np.array_equal(engine.row_checksums(P)[1:], st["csP"][1:])
    assert np.array_equal(H[-1].cpu().numpy(), st["bottom"])
    g = gran.cpu().numpy()
    assert np.array_equal(g >> 32, np.full(COLS + 1, 5)) and np.array_equal((g & 0xffffffff).astype(np.int32), st["bottom"])
    assert (done.cpu().numpy() == 5).all()


def test_config4_whole_matrix_p_only_monolithic_vs_eight_stacked_bands(engine, swamd, band0):
    import torch
    if _free_gb() < 150:
        pytest.skip(f"needs 150 GB of free HBM, {_free_gb()} GB free")
    a, b, st = band0
    n = COLS
    # (1) one launch
    d_a, _ = engine.to_device(a)
    d_b, _ = engine.to_device(b)
    out = engine.alloc(n, n, p_dtype=torch.int8, want_h=False)
    engine.fill_into(out, d_a, d_b)
    engine.synchronize()
    mono = out.result()
    cs_mono = engine.row_checksums(out.P)
    assert np.array_equal(cs_mono[1:BAND + 1], st["csP"][1:]), "rows 1..32768 against the oracle"
    # (2) eight band-resident launches, halo rows relayed chunk by chunk while they run
    m = swamd.MultiFill([0] * 8, a, b, p_dtype="int8", want_h=False)
    try:
        r = m.fill(nchunks=64)
        assert (r["max_score"], r["max_pos"]) == (mono["max_score"], mono["max_pos"])
        bands = m.band_tensors()
        assert [(lo, hi) for _, lo, hi, _, _ in bands] == [(k * BAND, (k + 1) * BAND) for k in range(8)]
        for _, lo, hi, _, Pb in bands:
            assert np.array_equal(engine.row_checksums(Pb)[1:], cs_mono[lo + 1:hi + 1]), f"band rows {lo + 1}..{hi}"
        plen_multi = m.traceback()
        plen_mono = engine.traceback(out, want_path=False)
        assert plen_multi == plen_mono and n < plen_mono < 3 * n
        # the negated paths agree too
        cs_mono = engine.row_checksums(out.P)
        for _, lo, hi, _, Pb in m.band_tensors():
            assert np.array_equal(engine.row_checksums(Pb)[1:], cs_mono[lo + 1:hi + 1]), f"band rows {lo + 1}..{hi} after the traceback"
    finally:
        m.close()
    del out
    torch.cuda.empty_cache()
