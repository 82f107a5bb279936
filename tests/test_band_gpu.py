"""GPU: the band-resident launch (sw_fill_band_device) -- the multi-GPU row-band decomposition with the halo row
travelling as {tag, H} granules while the kernels run -- on ONE GPU: stacked bands, in dependency order and with the
lower band launched FIRST on another stream (it must wait for its halo inside the kernel)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _bands(engine, oracle, swamd, cols, rows, cuts, reverse=False, p8=False, want_h=True, max_blocks=0):
    import torch
    a, b = oracle.generate(cols, rows, 41)
    H, P, mp = oracle.fill(a, b)
    bounds = [0] + list(cuts) + [rows]
    nb = len(bounds) - 1
    S = (cols + 62) // 63
    # concurrent launches need a context each (a context owns one arg-max key / edge workspace)
    engs = [swamd.Engine(0) for _ in range(nb)] if reverse else [engine] * nb
    d_a, _ = engine.to_device(a)
    gran = [torch.zeros(cols + 1, dtype=torch.int64, device="cuda") for _ in range(nb)]     # gran[g]: last row of band g
    done = [torch.zeros(S, dtype=torch.int32, device="cuda") for _ in range(nb)]
    streams = [torch.cuda.Stream() for _ in range(nb)]
    bands = []
    for g in range(nb):   # every allocation and copy first: nothing but the launches is enqueued while kernels wait
        lo, hi = bounds[g], bounds[g + 1]
        d_b, _ = engine.to_device(b[lo:hi])
        Hb = torch.zeros((hi - lo + 1, cols + 1), dtype=torch.int32, device="cuda") if want_h else None
        Pb = torch.zeros((hi - lo + 1, cols + 1), dtype=torch.int8 if p8 else torch.int32, device="cuda")
        res = torch.zeros(3, dtype=torch.int64, device="cuda")
        bands.append((Hb, Pb, res, lo, hi, d_b))
    if reverse:   # size the per-context workspaces up front (allocation inside a launch call would wait for the device)
        for g in range(nb):
            Hb, Pb, res, lo, hi, d_b = bands[g]
            engs[g].fill_band(d_a, cols, d_b, hi - lo, rows, None, None, res)
    torch.cuda.synchronize()
    for e in set(engs):
        e.set_option("max_blocks", max_blocks)
        e.set_option("band_wait_ms", 5000)
    try:
        for g in (reversed(range(nb)) if reverse else range(nb)):
            Hb, Pb, res, lo, hi, d_b = bands[g]
            with torch.cuda.stream(streams[g]):
                engs[g].fill_band(d_a, cols, d_b, hi - lo, rows, Hb, Pb, res,
                                  top_gran=gran[g - 1] if g > 0 else None, top_tag=7 + g - 1 if g > 0 else 0,
                                  bot_gran=gran[g], bot_tag=7 + g, bot_done=done[g], concurrent=reverse)
        torch.cuda.synchronize()
    finally:
        for e in set(engs):
            e.set_option("max_blocks", 0)
            e.set_option("band_wait_ms", 0)
    best = (0, 0)
    for g in range(nb):
        Hb, Pb, res, lo, hi, _ = bands[g]
        r = res.cpu().tolist()
        assert r[2] == 0, f"band {g} aborted"
        if want_h:
            assert np.array_equal(Hb.cpu().numpy(), H[lo:hi + 1]), f"band {g} H"
        assert np.array_equal(Pb.cpu().numpy()[1:].astype(np.int32), P[lo + 1:hi + 1]), f"band {g} P"
        gr = gran[g].cpu().numpy()
        assert np.array_equal(gr >> 32, np.full(cols + 1, 7 + g)) and np.array_equal((gr & 0xffffffff).astype(np.int32), H[hi]), f"band {g} bottom granules"
        assert (done[g].cpu().numpy() == 7 + g).all()
        if r[1] > 0:
            gpos = r[0] + lo * (cols + 1)
            if r[1] > best[0] or (r[1] == best[0] and gpos < best[1]):
                best = (r[1], gpos)
    assert best == (int(H.flat[mp]), mp)
    if reverse:
        for e in engs:
            e.close()


@pytest.mark.parametrize("cols,rows,cuts", [(777, 400, (160,)), (4200, 1300, (416, 880)), (130, 200, (16, 32, 48))])
def test_stacked_bands_in_order(engine, oracle, swamd, cols, rows, cuts):
    _bands(engine, oracle, swamd, cols, rows, cuts)


def test_stacked_bands_compact_p_no_h(engine, oracle, swamd):
    _bands(engine, oracle, swamd, 1500, 900, (304, 608), p8=True, want_h=False)


def test_lower_band_launched_first_waits_for_its_halo(engine, oracle, swamd):
    """Band 1 is enqueued BEFORE band 0, each on its own stream with half of the CUs: band 1's strips poll their halo
    granules inside the kernel until band 0, running beside it, has produced them."""
    ncu = engine.get_option("num_cus")
    _bands(engine, oracle, swamd, 5000, 2000, (992,), reverse=True, max_blocks=max(8, ncu // 2 - 8))
