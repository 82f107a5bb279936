"""GPU: BASELINE config 4 (262144 x 262144, row bands over 8 GPUs) AT SIZE on one MI355X, against WHOLE-MATRIX digests of the
streaming oracle (tests/golden/big_digests.json, made once in the build container by tests/golden/make_big_digests.py: the oracle's
6.9e10-cell fill takes 12 minutes on one core -- the GPU box no longer repeats it).

  * band 0 of the 8-way split at full width -- 262144 columns x 32768 rows, int32 H + int8 P, one band-resident launch
    (sw_fill_band_device): per-row checksums of H and P, arg-max, the bottom row both as matrix row and as the granules the next
    band would poll;
  * the WHOLE 262144^2 matrix as int8 P only (69 GB: what one GPU can keep), two ways -- one monolithic launch and eight
    stacked band-resident launches through sw_multi_* (devices [0]*8: the relay, the chunking and the tags of the 8-GPU run,
    the bands sharing this GPU's CUs): every row of P (all 262144, band by band), the arg-max, the traced-back path (length, every
    index, where it ends) and P after the traceback, each against the oracle's digest."""
import json
import os

import numpy as np
import pytest

from oracle_lib import GOLDEN

pytestmark = pytest.mark.gpu
COLS, BAND = 262144, 32768
DIGEST = json.load(open(os.path.join(GOLDEN, "big_digests.json")))["rand_262144x262144_s1"]


def fnv(oracle, arr):
    return f"{oracle.fnv(np.ascontiguousarray(arr)):016x}"


@pytest.fixture(scope="module")
def seqs(swamd):
    return swamd.generate(COLS, COLS, 1)


def _free_gb():
    import torch
    return torch.cuda.mem_get_info()[0] >> 30


def test_band0_of_config4_full_width_vs_oracle_digest(engine, oracle, seqs):
    import torch
    if _free_gb() < 60:
        pytest.skip(f"needs 60 GB of free HBM, {_free_gb()} GB free")
    a, b = seqs
    d0 = DIGEST["bands"][0]
    assert (d0["lo"], d0["hi"]) == (0, BAND)
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
    assert (r[0], r[1]) == (d0["maxPos"], d0["maxScore"])
    csH, csP = engine.row_checksums(H), engine.row_checksums(P)
    assert csH[0] == 0 and fnv(oracle, csH[1:]) == d0["fnv_csH"]
    assert fnv(oracle, csP[1:]) == d0["fnv_csP"]
    bottom = H[-1].cpu().numpy()
    assert fnv(oracle, bottom) == d0["fnv_bottom_H"]
    g = gran.cpu().numpy()
    assert np.array_equal(g >> 32, np.full(COLS + 1, 5)) and np.array_equal((g & 0xffffffff).astype(np.int32), bottom)
    assert (done.cpu().numpy() == 5).all()


def test_config4_whole_matrix_p_only_monolithic_vs_eight_stacked_bands(engine, oracle, swamd, seqs):
    import torch
    if _free_gb() < 150:
        pytest.skip(f"needs 150 GB of free HBM, {_free_gb()} GB free")
    a, b = seqs
    n = COLS
    bands = DIGEST["bands"]

    def check_rows(cs_of_band, key, what):
        for d in bands:
            assert fnv(oracle, cs_of_band(d["lo"], d["hi"])) == d[key], f"{what}: rows {d['lo'] + 1}..{d['hi']} differ from the oracle"

    # (1) one launch
    d_a, _ = engine.to_device(a)
    d_b, _ = engine.to_device(b)
    out = engine.alloc(n, n, p_dtype=torch.int8, want_h=False)
    engine.fill_into(out, d_a, d_b)
    engine.synchronize()
    mono = out.result()
    assert (mono["max_pos"], mono["max_score"]) == (DIGEST["maxPos"], DIGEST["maxScore"])
    cs_mono = engine.row_checksums(out.P)
    assert fnv(oracle, cs_mono) == DIGEST["fnv_csP"], "whole matrix"
    check_rows(lambda lo, hi: cs_mono[lo + 1:hi + 1], "fnv_csP", "monolithic")
    # (2) eight band-resident launches, halo rows relayed chunk by chunk while they run
    m = swamd.MultiFill([0] * 8, a, b, p_dtype="int8", want_h=False)
    try:
        r = m.fill(nchunks=64)
        assert (r["max_score"], r["max_pos"]) == (DIGEST["maxScore"], DIGEST["maxPos"])
        bt = m.band_tensors()
        assert [(lo, hi) for _, lo, hi, _, _ in bt] == [(d["lo"], d["hi"]) for d in bands]
        cs_b = {lo: engine.row_checksums(Pb) for _, lo, hi, _, Pb in bt}
        check_rows(lambda lo, hi: cs_b[lo][1:], "fnv_csP", "stacked bands")
        plen_multi = m.traceback()
        path = engine.traceback(out)
        assert plen_multi == len(path) == DIGEST["pathLen"]
        assert fnv(oracle, path) == DIGEST["fnv_path"] and int(path[-1]) == DIGEST["path_end"]
        # the negated paths agree too
        cs_mono = engine.row_checksums(out.P)
        assert fnv(oracle, cs_mono) == DIGEST["fnv_csP1"]
        cs_b = {lo: engine.row_checksums(Pb) for _, lo, hi, _, Pb in m.band_tensors()}
        check_rows(lambda lo, hi: cs_b[lo][1:], "fnv_csP1", "stacked bands after the traceback")
    finally:
        m.close()
    del out
    torch.cuda.empty_cache()
