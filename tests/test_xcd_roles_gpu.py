"""GPU: the two-column kernel's per-XCD role assignment (sw_systolic2.inc, xcd_mode) -- every XCD runs an eighth of the
fillers and the scouts that feed them, and a scout keeps its edge stores in that XCD's L2 (" sc0") when it has seen that
both readers of its edge column really run there.  Same results as the oracle and as the layout without it (debug bit 23),
across the strip counts where the layout changes shape: 16 strips (the smallest), remainders over 8, the first counts with
two-strip scout workgroups (fillers + scouts > 32 per XCD) and the largest count that still has scouts."""
import numpy as np
import pytest

from test_fill_gpu import check_against_oracle

pytestmark = pytest.mark.gpu

XCD_OFF = 8388608


def _needs_round_robin(engine):
    if engine.get_option("xcd_round_robin") != 1:
        pytest.skip("workgroup i is not on XCD i % 8 on this device (sw_xcc_probe): the per-XCD layout is not used")


@pytest.mark.parametrize("strips", [16, 17, 23, 24, 64, 127, 128, 129, 131, 137, 160, 167])
def test_xcd_layout_matches_oracle_and_plain_layout(engine, oracle, strips):
    _needs_round_robin(engine)
    cols, rows = 126 * strips - 5, 272
    a, b = oracle.generate(cols, rows, 300 + strips)
    check_against_oracle(engine, oracle, a, b)
    assert engine.get_option("last_strips2") == strips and engine.get_option("last_scouts") > 0
    assert engine.get_option("last_xcd_mode") == 1, "the per-XCD layout was expected to run"
    with_xcd = engine.fill(a, b)
    engine.set_option("debug_flags", XCD_OFF)
    try:
        plain = engine.fill(a, b)
        assert engine.get_option("last_xcd_mode") == 0
    finally:
        engine.set_option("debug_flags", 0)
    assert with_xcd.result() == plain.result()
    assert np.array_equal(with_xcd.H.cpu().numpy(), plain.H.cpu().numpy()) and np.array_equal(with_xcd.P.cpu().numpy(), plain.P.cpu().numpy())


def test_xcd_layout_limits(engine, oracle):
    """below 16 strips and where the scouts no longer fit (more than 1.5 x 168 workgroups) the plain layouts run"""
    _needs_round_robin(engine)
    for strips, want in ((15, 0), (16, 1), (167, 1)):
        a, b = oracle.generate(126 * strips, 64, 7)
        check_against_oracle(engine, oracle, a, b)
        assert engine.get_option("last_xcd_mode") == want, f"{strips} strips"
    # 200 strips of 126 columns: two column tiles of 100 strips, each dealt per XCD; untiled (debug bit 19) the plain classic chain
    a, b = oracle.generate(126 * 200, 64, 7)
    engine.set_option("s2w", 126)
    try:
        check_against_oracle(engine, oracle, a, b)
        assert engine.get_option("last_tiles") == 2 and engine.get_option("last_xcd_mode") == 1
        engine.set_option("debug_flags", 524288)
        check_against_oracle(engine, oracle, a, b)
        assert engine.get_option("last_tiles") == 1 and engine.get_option("last_xcd_mode") == 0
    finally:
        engine.set_option("debug_flags", 0); engine.set_option("s2w", 0)
    # the library's own choice there: one launch of overlapping strips (229 of them), no scouts
    check_against_oracle(engine, oracle, a, b)
    assert engine.get_option("last_tiles") == 1 and engine.get_option("last_strips2") == -(-(126 * 200 - 126) // 110) + 1


@pytest.mark.parametrize("mode", ["p8", "h64", "p8_only", "score_only"])
def test_xcd_layout_other_formats(engine, oracle, mode):
    import torch
    _needs_round_robin(engine)
    a, b = oracle.generate(126 * 40, 333, 84)
    H, P, mp = oracle.fill(a, b)
    out = engine.fill(a, b, h_dtype=torch.int64 if mode == "h64" else None, p_dtype=torch.int8 if mode.startswith("p8") else None,
                      want_h=mode in ("p8", "h64"), want_p=mode != "score_only")
    if engine.get_option("last_strips2") == 40 and engine.get_option("last_scouts") > 0:
        assert engine.get_option("last_xcd_mode") == 1
    r = out.result()
    assert r["max_pos"] == mp and r["max_score"] == int(H.flat[mp])
    if out.H is not None:
        assert np.array_equal(out.H.cpu().numpy().astype(np.int32), H)
    if out.P is not None:
        assert np.array_equal(out.P.cpu().numpy().astype(np.int32), P)


def test_repeated_fills_of_changing_sizes(engine, oracle):
    """the roles (and which XCD writes which edge column) change with the strip count from launch to launch: nothing of an
    earlier launch may be taken for this one's edge values (self-tagged) or placement table (cleared by the preparation kernel)"""
    _needs_round_robin(engine)
    for k, strips in enumerate([131, 17, 64, 131, 16, 129, 40, 131]):
        a, b = oracle.generate(126 * strips - k, 144 + 16 * k, 500 + k)
        check_against_oracle(engine, oracle, a, b)
        assert engine.get_option("last_xcd_mode") == 1


@pytest.mark.parametrize("strips,max_blocks,rows", [(70, 0, 272), (200, 64, 144), (200, 100, 80), (133, 0, 333)])
def test_classic_chain_dealt_per_xcd(engine, oracle, strips, max_blocks, rows):
    """xcd_mode 2: no scouts (debug bit 17), the strips of a pass dealt per XCD, edge columns kept in the XCD's L2 except across the
    seams; one pass, several passes (max_blocks), a last pass that is not full"""
    _needs_round_robin(engine)
    a, b = oracle.generate(126 * strips - 3, rows, 900 + strips)
    engine.set_option("debug_flags", 131072)
    engine.set_option("xcd_chain", 1)
    engine.set_option("max_blocks", max_blocks)
    try:
        check_against_oracle(engine, oracle, a, b)
        assert engine.get_option("last_scouts") == 0 and engine.get_option("last_xcd_mode") == 2
        engine.set_option("xcd_chain", 2)
        check_against_oracle(engine, oracle, a, b)
        assert engine.get_option("last_xcd_mode") == 0
    finally:
        engine.set_option("debug_flags", 0); engine.set_option("xcd_chain", 0); engine.set_option("max_blocks", 0)
