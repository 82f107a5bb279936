"""GPU: sw_multi_* -- one matrix over several GPUs of one process, behind the C-ABI.  On the one-GPU test box the device
list repeats GPU 0 (the bands then share its CUs, halo chunks relayed with device-to-device copies while the kernels
run); with more GPUs visible the same test also runs over distinct devices (peer copies over xGMI)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _check(swamd, oracle, devices, cols, rows, p_dtype, nchunks):
    a, b = oracle.generate(cols, rows, 9)
    H, P, mp = oracle.fill(a, b)
    m = swamd.MultiFill(devices, a, b, p_dtype=p_dtype)
    try:
        for _ in range(2):
            r = m.fill(nchunks=nchunks)
        assert (r["max_score"], r["max_pos"]) == (int(H.flat[mp]), mp)
        plen = m.traceback()
        path = oracle.backtrack(P, mp)   # P negated along the path
        assert plen == len(path)
        for dev, lo, hi, bH, bP in m.bands():
            assert np.array_equal(bH[1:], H[lo + 1:hi + 1]), f"band rows {lo}..{hi} H"
            assert np.array_equal(bP[1:], P[lo + 1:hi + 1]), f"band rows {lo}..{hi} P (incl. negated path)"
    finally:
        m.close()


@pytest.mark.parametrize("devices,cols,rows,p_dtype,nchunks", [([0], 1500, 700, "int32", 8), ([0, 0], 3000, 1200, "int32", 8),
                                                               ([0, 0, 0], 2000, 1000, "int8", 64), ([0, 0], 100, 40, "int32", 4)])
def test_bands_sharing_one_gpu(swamd, oracle, devices, cols, rows, p_dtype, nchunks):
    _check(swamd, oracle, devices, cols, rows, p_dtype, nchunks)


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (peer copies of the halo rows)")
def test_bands_on_two_gpus(swamd, oracle):
    _check(swamd, oracle, [0, 1], 20000, 6000, "int8", 64)
