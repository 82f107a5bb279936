"""GPU: sw_align_auto_multi -- the adaptive dispatch with the reference's three executors (omp_smithW-v7-adaptive.cpp:304-396:
serial / OpenMP / offload) as host / one GPU / row bands over several GPUs.  Whatever leg runs, H, P (after the traceback), maxPos,
score and path length are the reference's."""
import numpy as np
import pytest

from oracle_lib import golden

pytestmark = pytest.mark.gpu


def check(r, oracle, a, b):
    H, P, mp = oracle.fill(a, b)
    path = oracle.backtrack(P, mp)
    assert np.array_equal(r["H"], H) and np.array_equal(r["P"], P)
    assert r["max_pos"] == mp and r["max_score"] == int(H.flat[mp]) and r["path_len"] == len(path)


def test_three_executors_by_size(engine, swamd, oracle):
    a, b = swamd.generate(300, 200, 1)                     # 6e4 cells: the host fill
    r = swamd.align_auto(a, b, engine=engine, devices=[0, 0])
    assert r["executor"] == 0 and not r["used_gpu"]
    check(r, oracle, a, b)
    a, b = swamd.generate(2100, 1300, 3)                   # one GPU: a single pair this small is chain-bound
    r = swamd.align_auto(a, b, engine=engine, devices=[0, 0])
    assert r["executor"] == 1
    check(r, oracle, a, b)
    r = swamd.align_auto(a, b, engine=None, devices=[0])   # (no context given: one is made on devices[0])
    assert r["executor"] == 1
    check(r, oracle, a, b)
    r = swamd.align_auto(a, b, engine=engine, devices=[0, 0], multi_min_cells=1000000)   # the N-GPU leg: two bands (here on one GPU)
    assert r["executor"] == 2
    check(r, oracle, a, b)
    r = swamd.align_auto(a, b, engine=engine, devices=[0, 0, 0], multi_min_cells=1)
    assert r["executor"] == 2
    check(r, oracle, a, b)


def test_multi_leg_matches_reference_fixture(swamd):
    g = golden("rand_300x200_s1")
    r = swamd.align_auto(g["a"], g["b"], devices=[0, 0], multi_min_cells=1)   # (6e4 cells stay on the host whatever the threshold)
    assert r["executor"] == 0
    assert np.array_equal(r["H"], g["H"]) and np.array_equal(r["P"], g["P1"])


def test_multi_leg_on_two_devices(swamd, oracle):
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    a, b = swamd.generate(5000, 3000, 2)
    r = swamd.align_auto(a, b, devices=[0, 1], multi_min_cells=1)
    assert r["executor"] == 2
    check(r, oracle, a, b)
