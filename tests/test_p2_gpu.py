"""GPU: the 2-bit predecessor matrix: pack / unpack round trip to the reference's int32 layout and backtrack()
(serial_smithW.c:262-277) on the packed form, against the reference fixtures and the oracle."""
import numpy as np
import pytest

from oracle_lib import golden

pytestmark = pytest.mark.gpu


def np_pack(P):
    """The format by definition: cell k in bits 2 (k & 3) of byte k >> 2; path bitmap bit k & 31 of word k >> 5."""
    flat = P.reshape(-1).astype(np.int64)
    n = flat.size
    pad = (-n) % 32
    codes = np.concatenate([np.abs(flat), np.zeros(pad, np.int64)]).astype(np.uint8).reshape(-1, 4)
    p2 = (codes[:, 0] | (codes[:, 1] << 2) | (codes[:, 2] << 4) | (codes[:, 3] << 6)).astype(np.uint8)
    neg = np.concatenate([flat < 0, np.zeros(pad, bool)]).reshape(-1, 32)
    bits = (neg.astype(np.uint64) << np.arange(32, dtype=np.uint64)).sum(axis=1).astype(np.uint32)
    return p2, bits


@pytest.mark.parametrize("name", ["kat_builtin", "rand_1x1_s1", "rand_8x9_s1", "rand_256x256_s1", "rand_300x200_s1", "rand_65x130_s7", "rand_1x77_s3",
                                  "rand_77x1_s3", "rand_129x64_s11"])
@pytest.mark.parametrize("src", ["int32", "int8"])
def test_p2_round_trip_and_traceback_vs_reference(engine, name, src):
    import torch
    g = golden(name)
    P0, P1, path, mp = g["P0"], g["P1"], g["path"], int(g["meta"][3])
    rows1, m = P0.shape
    dt = torch.int32 if src == "int32" else torch.int8
    dP0 = torch.from_numpy(P0).to("cuda").to(dt)
    P2, bits = engine.pack_p2(dP0)
    engine.synchronize()
    want2, wantb = np_pack(P0)
    assert np.array_equal(P2.cpu().numpy(), want2) and not bits.cpu().numpy().any()
    assert np.array_equal(engine.unpack_p2(P2, None, P0.shape).cpu().numpy(), P0)
    # backtrack() on the packed matrix: same path, and the bitmap + codes give the reference's negated P back
    bits.zero_()
    got = engine.traceback_p2(P2, m - 1, rows1 - 1, mp, bits)
    assert np.array_equal(got, path)
    assert np.array_equal(engine.unpack_p2(P2, bits, P0.shape).cpu().numpy(), P1)
    assert np.array_equal(P2.cpu().numpy(), want2), "the walk must not modify the codes"
    # packing an already traced matrix recovers the path as the bitmap
    P2b, bitsb = engine.pack_p2(torch.from_numpy(P1).to("cuda").to(dt))
    engine.synchronize()
    w2, wb = np_pack(P1)
    assert np.array_equal(P2b.cpu().numpy(), w2) and np.array_equal(bitsb.cpu().numpy().view(np.uint32), wb)
    assert np.array_equal(bitsb.cpu().numpy(), bits.cpu().numpy())


def test_p2_of_a_device_filled_matrix_4096(engine, oracle, swamd):
    """int8 P of a 4096 x 4096 fill -> 2 bits per cell (4.2 MB instead of 16.8 MB / 67 MB): the packed walk visits the cells the int8 walk
    and the oracle visit; unpacked it is the oracle's P after backtrack()."""
    import torch
    a, b = swamd.generate(4096, 4096, 1)
    out = engine.fill(a, b, p_dtype=torch.int8)
    r = out.result()
    P2, bits = engine.pack_p2(out.P)
    bits.zero_()
    assert P2.numel() * 4 >= out.P.numel() and P2.numel() <= out.P.numel() // 4 + 8
    path2 = engine.traceback_p2(P2, 4096, 4096, r["max_pos"], bits)
    path8 = engine.traceback(out, r["max_pos"])
    assert np.array_equal(path2, path8)
    H, P, mp = oracle.fill(a, b)
    assert mp == r["max_pos"] and np.array_equal(oracle.backtrack(P, mp), path2)
    assert np.array_equal(engine.unpack_p2(P2, bits, out.P.shape).cpu().numpy(), P)
    assert np.array_equal(out.P.cpu().numpy().astype(np.int32), P)


def test_p2_argument_errors(engine, swamd):
    L = swamd.lib()
    assert L.sw_p_to_p2_device(engine._h, None, 1, None, None, 10, None) == -22
    assert L.sw_p_to_p2_device(engine._h, 16, 2, 16, None, 10, None) == -22
    assert L.sw_p2_to_p32_device(engine._h, None, None, None, 10, None) == -22
    assert L.sw_traceback_p2_device(engine._h, 16, 4, 4, 25, None, None, 0, 16, None) == -22
    assert L.sw_p_to_p2_device(engine._h, None, 1, None, None, 0, None) == 0
